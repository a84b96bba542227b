"""RCCL called directly (ctypes over the librccl.so that ships with PyTorch-ROCm), so that the ONE
collective of the hot path -- the all-gather of the predicted betas -- runs on a stream WE choose.

Why not ``torch.distributed`` for it: c10d's NCCL backend runs every collective on a stream of its
own.  HIP multiplexes a process's streams onto a few hardware queues, and the four-lane backbone is
sensitive to every extra one: with a world-size-1 RCCL group on ONE GPU the step was 17 % slower
(4,140-4,170 vs 5,020 images/s, both issue modes of ``BetasGatherer``; profiles/r04j_*), before a
single byte crosses xGMI.  Here the collective is enqueued on a stream the process already has: the
caller's compute stream, or one of the executor's branch lanes (BetasGatherer mode 'lane').

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
    CURRENT device.  ``all_gather(local)`` enqueues on the current stream (or on ``stream``) and
    returns the gathered tensor (valid in stream order, like any kernel output).

    Construction is split so that a caller can agree with the other ranks BEFORE anybody enters a
    collective (shapy_amd/parallel.py: BetasGatherer._init_rccl):
        c = RcclComm.prepare(group=g)    # local only: loads librccl, rank 0 of the group draws the id
        ...                              # all ranks agree that every prepare() succeeded
        c.connect()                      # collective: id broadcast + blocking ncclCommInitRank
    ``RcclComm(rank, world, group)`` does both (single-rank use, tests)."""

    def __init__(self, rank=None, world=None, group=None, _connect=True):
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        lib = _load()
        if dist.is_initialized():
            rank = dist.get_rank(group) if rank is None else rank
            world = dist.get_world_size(group) if world is None else world
        else:
            rank, world = rank or 0, world or 1
        self.rank, self.world, self.group, self._lib = rank, world, group, lib
        self._comm = ctypes.c_void_p()
        self.device = None
        self._uid = _UniqueId()
        if rank == 0:                          # (the GROUP's rank 0)
            _check(lib, lib.ncclGetUniqueId(ctypes.byref(self._uid)), 'ncclGetUniqueId')
        if _connect:
            self.connect()

    @classmethod
    def prepare(cls, rank=None, world=None, group=None):
        return cls(rank, world, group, _connect=False)

    def connect(self):
        """Collective over the group: every rank must call it (after agreeing that all of them can)."""
        lib = self._lib
        if self.world > 1:
            box = [bytes(self._uid.internal) if self.rank == 0 else None]
            # src is a GLOBAL rank: the group's rank 0 need not be the world's
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast_object_list(box, src=src, group=self.group)
            ctypes.memmove(ctypes.byref(self._uid), box[0], 128)
        _check(lib, lib.ncclCommInitRank(ctypes.byref(self._comm), self.world, self._uid, self.rank),
               'ncclCommInitRank')
        self.device = torch.cuda.current_device() if torch.cuda.is_available() else None
        return self

    def all_gather(self, local, stream=None, out=None):
        """stream: raw hipStream_t value to enqueue on (default: torch's current stream)."""
        if not (local.is_cuda and local.dtype == torch.float32):
            raise ValueError('RcclComm.all_gather: float32 CUDA tensors')
        local = local.contiguous()
        if out is None:
            out = local.new_empty((self.world * local.shape[0],) + tuple(local.shape[1:]))
        if stream is None:
            stream = torch.cuda.current_stream(local.device).cuda_stream
        _check(self._lib, self._lib.ncclAllGather(local.data_ptr(), out.data_ptr(), local.numel(),
                                                  _NCCL_FLOAT, self._comm, ctypes.c_void_p(stream)),
               'ncclAllGather')
        return out

    def close(self):
        if self._comm:
            if self.device is not None:
                torch.cuda.synchronize(self.device)
            self._lib.ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:            # interpreter shutdown
            pass
