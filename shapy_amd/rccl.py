"""RCCL called directly (ctypes over the librccl.so that ships with PyTorch-ROCm), so that the ONE
collective of the hot path -- the all-gather of the predicted betas -- runs on a stream WE choose.

Why not ``torch.distributed`` for it: c10d's NCCL backend runs every collective on a stream of its
own.  HIP multiplexes a process's streams onto a few hardware queues, and the four-lane backbone is
sensitive to every extra one: with a world-size-1 RCCL group on ONE GPU the step was 17 % slower
(4,140-4,170 vs 5,020 images/s, both issue modes of ``BetasGatherer``; profiles/r04j_*), before a
single byte crosses xGMI.  Here the collective is enqueued on the caller's compute stream, behind
the tail of the step: no stream, no event, ~tens of microseconds of stream time per step.

The control plane (rendezvous, barriers, the max-over-ranks of the timing) stays with
``torch.distributed`` -- any backend; only the 128-byte ``ncclUniqueId`` travels over it.
"""
import ctypes
import os
import os.path as osp

import torch
import torch.distributed as dist

_NCCL_FLOAT = 7          # ncclFloat32 (nccl.h: ncclDataType_t)
_lib = None


class _UniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_byte * 128)]


class RcclError(RuntimeError):
    pass


def _load():
    global _lib
    if _lib is not None:
        return _lib
    path = osp.join(osp.dirname(torch.__file__), 'lib', 'librccl.so')
    if not osp.exists(path):
        raise RcclError(f'librccl.so not found next to torch ({path})')
    lib = ctypes.CDLL(path)                 # the instance torch itself has mapped
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [i32]
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), i32, _UniqueId, i32]     # id BY VALUE
    lib.ncclAllGather.argtypes = [vp, vp, ctypes.c_size_t, i32, vp, vp]
    lib.ncclCommDestroy.argtypes = [vp]
    for f in ('ncclGetUniqueId', 'ncclCommInitRank', 'ncclAllGather', 'ncclCommDestroy'):
        getattr(lib, f).restype = i32
    _lib = lib
    return lib


def _check(lib, rc, what):
    if rc != 0:
        raise RcclError(f'{what}: {lib.ncclGetErrorString(rc).decode()} ({rc})')


class RcclComm:
    """One RCCL communicator over the ranks of ``group`` (default: the world), this process's
    CURRENT device.  ``all_gather(local)`` enqueues on the current stream and returns the gathered
    tensor (valid in stream order, like any kernel output)."""

    def __init__(self, rank=None, world=None, group=None):
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        lib = _load()
        if dist.is_initialized():
            rank = dist.get_rank(group) if rank is None else rank
            world = dist.get_world_size(group) if world is None else world
        else:
            rank, world = rank or 0, world or 1
        self.rank, self.world, self._lib = rank, world, lib
        uid = _UniqueId()
        if rank == 0:
            _check(lib, lib.ncclGetUniqueId(ctypes.byref(uid)), 'ncclGetUniqueId')
        if world > 1:
            box = [bytes(uid.internal) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            ctypes.memmove(ctypes.byref(uid), box[0], 128)
        self._comm = ctypes.c_void_p()
        _check(lib, lib.ncclCommInitRank(ctypes.byref(self._comm), world, uid, rank), 'ncclCommInitRank')
        self.device = torch.cuda.current_device()

    def all_gather(self, local):
        if not (local.is_cuda and local.dtype == torch.float32):
            raise ValueError('RcclComm.all_gather: float32 CUDA tensors')
        local = local.contiguous()
        out = local.new_empty((self.world * local.shape[0],) + tuple(local.shape[1:]))
        stream = torch.cuda.current_stream(local.device).cuda_stream
        _check(self._lib, self._lib.ncclAllGather(local.data_ptr(), out.data_ptr(), local.numel(),
                                                  _NCCL_FLOAT, self._comm, ctypes.c_void_p(stream)),
               'ncclAllGather')
        return out

    def close(self):
        if self._comm:
            torch.cuda.synchronize(self.device)
            self._lib.ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:            # interpreter shutdown
            pass
