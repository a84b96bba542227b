"""Data parallelism of the hot path over the GPUs of one node.

Images are independent, BatchNorm is in eval mode and the weights (418 MB) and SMPL-X
buffers (65 MB) are replicated, so the forward pass needs NO exchange: every rank (one
process per GPU) runs a contiguous shard of the batch.  The only collective is the
all-gather of the predicted betas ([B_local, 10] float32 = 1,280 B per rank at bs=256/8)
at the end of a step -- latency-bound, so it is issued as ONE asynchronous RCCL all_gather and
joined into the compute stream only when the NEXT step issues its own gather
(``BetasGatherer``): it overlaps the next batch's backbone.

The reference has no data-parallel inference at all (rank > 0 returns immediately in its
evaluator, regressor/human_shape/evaluation.py:641-642; the only collective it executes is
a barrier, regressor/evaluate.py:100-105) -- this is new functionality asked for by the
north star, shaped for xGMI: one tiny collective per step, nothing bucketed.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous split of n items: the first n % world ranks get one extra item."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_distributed(backend='nccl'):
    """env:// rendezvous as in regressor/evaluate.py:68-79 (backend "nccl" is RCCL on ROCm)."""
    import os
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if not dist.is_initialized():
        dist.init_process_group(backend, init_method='env://')
    return dist.get_rank(), dist.get_world_size()


class BetasGatherer:
    """all_gather of equally sized per-rank tensors, asynchronous and DEFERRED: the current
    stream does not wait for the collective when it is issued, so the (latency-bound) RCCL
    call overlaps whatever the caller enqueues next -- in ``bench.py`` the backbone of the
    next batch.

        g = BetasGatherer(world)
        for batch in batches:
            prev = g(betas_of(batch))     # issues this step's gather; the tensor returned by
                                          # the PREVIOUS call is complete from here on
        last = g.wait()                   # joins the last gather into the current stream

    ``gather(local)`` = issue + wait for callers that need the result right away.  On CPU
    tensors (gloo) the collective is synchronous and every form returns a finished tensor."""

    def __init__(self, world=None, group=None, force=False, mode=None):
        """force: take the collective path even for ONE rank (bench.py --force-gather: a world-size-1
        RCCL group exercises the stream / event structure of the N-rank path on a single GPU).
        mode: 'lane' (default) = ncclAllGather called directly (shapy_amd/rccl.py) on the executor's
        lane-1 stream -- a stream the process already has (c10d's own RCCL stream alone cost the
        four-lane backbone 17 % on one GPU, 4,140-4,170 vs 5,020 images/s, profiles/r04j_*: HIP
        multiplexes a process's streams onto a few hardware queues) -- behind an event recorded on the
        caller's stream once the betas exist, and joined back into the caller's stream by the NEXT
        call.  The compute stream never queues behind the collective, i.e. never behind the slowest
        rank: lane 1 gets its first op of the next forward ~1.4 ms into it (transition1), by which
        time the latency-bound gather has long finished.
        'work' = the fallback: torch.distributed's NCCL backend, async_op=True from the caller's stream,
        the Work handle joined at the next call (what every rank takes together when the direct
        communicator cannot be built; selectable for A/B runs).  The compute-stream and private-side-
        stream forms of rounds 1-4 lost on one GPU (profiles/r04j_*, r05c_*) and are gone.
        SHAPY_GATHER_MODE overrides the default.
        ``last_join_wait_ms`` (after ``wait(measure=True)``): how long the host found the lane's `done`
        event still pending -- at N > 1 the only evidence of a lane-1 stall behind a slow rank; bench.py
        reports its maximum over the timed steps."""
        import os
        self.group = group
        self.world = world if world is not None else (
            dist.get_world_size(group) if dist.is_initialized() else 1)
        self.force = bool(force)
        self.mode = mode or os.environ.get('SHAPY_GATHER_MODE', 'lane')
        if self.mode not in ('lane', 'work'):
            raise ValueError(f'unknown gather mode {self.mode!r}')
        self._comm = None             # mode 'lane': shapy_amd.rccl.RcclComm, created on first use
        self.last_join_wait_ms = 0.0
        self._lane = None             # mode 'lane': the executor's lane-1 stream (torch.cuda.ExternalStream)
        self._pending = None          # (out, event | Work | None) of the gather still in flight
        self.issued = 0
        self.deferred_waits = 0       # waits that were served by a LATER call (the overlap)

    def wait(self, measure=False):
        """Makes the current stream wait for the gather in flight (if any); returns its result.
        measure: the HOST also waits for the lane's `done` event and records how long that took
        (diagnostics only: it serialises the host with the device)."""
        if self._pending is None:
            return None
        out, h = self._pending[:2]
        self._pending = None
        if h is None:
            pass
        elif isinstance(h, torch.cuda.Event):
            if measure:
                import time
                t0 = time.perf_counter()
                h.synchronize()
                self.last_join_wait_ms = (time.perf_counter() - t0) * 1e3
            torch.cuda.current_stream().wait_event(h)
        else:
            h.wait()                  # c10d Work: the CURRENT STREAM waits for the RCCL stream
        return out

    def __call__(self, local):
        if self.world == 1 and not self.force:
            return local
        if self._pending is not None:
            self.deferred_waits += 1
            self.wait()
        local = local.contiguous()
        out = local.new_empty((self.world * local.shape[0],) + tuple(local.shape[1:]))
        self.issued += 1
        if local.is_cuda and self.mode == 'lane':
            if self._comm is None and not self._init_rccl():
                return self._call_c10d(local, out)          # every rank fell back to mode 'work'
            # on the executor's lane-1 stream, behind the producer of `local`; joined by the next call
            lane = self._lane_stream(local.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(local.device))
            lane.wait_event(ready)
            self._comm.all_gather(local, stream=lane.cuda_stream, out=out)
            done = torch.cuda.Event()
            done.record(lane)
            self._pending = (out, done, local)     # `local` / `out` stay referenced until the join
        elif local.is_cuda:
            return self._call_c10d(local, out)
        else:
            chunks = list(out.chunk(self.world, dim=0))
            dist.all_gather(chunks, local, group=self.group)
            self._pending = (out, None)
        return out

    def _lane_stream(self, device):
        if self._lane is None:
            import ctypes
            from . import _lib
            h = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.check(_lib.load().shapy_hrnet_lane_stream(1, ctypes.byref(h)), 'shapy_hrnet_lane_stream')
            self._lane = torch.cuda.ExternalStream(h.value, device=device)
        return self._lane

    def _agree(self, ok):
        """min over the ranks of an ok flag (control plane); one rank: the flag itself."""
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dev = 'cuda' if dist.get_backend(self.group) == 'nccl' else 'cpu'
            flag = torch.tensor([int(ok)], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return int(flag.item())
        return int(ok)

    def _init_rccl(self):
        """Creates the direct RCCL communicator on first use.  Returns False when ANY rank could not
        (agreed over the control plane, so that all ranks take the same path): the gatherer then
        falls back to c10d's NCCL backend (mode 'work'; a 'nccl' subgroup is created when the default
        process group is not one) -- slower per step (an extra stream), but a working collective."""
        import logging
        import os
        from .rccl import RcclComm
        # Two phases, each followed by an agreement over the control plane, so that NO rank enters a
        # collective of the communicator's construction unless every rank will: (1) local -- load
        # librccl, the group's rank 0 draws the unique id; (2) collective -- id broadcast + the blocking
        # ncclCommInitRank.  A rank that fails in (1) (librccl missing, ncclGetUniqueId error, the test
        # hook on a subset of ranks) makes everybody skip (2) and take the c10d fallback together.
        ok, err = 1, None
        try:
            force = os.environ.get('SHAPY_RCCL_FORCE_FAIL', '')
            if force == '1' or (force.startswith('rank') and dist.is_initialized()
                                and dist.get_rank(self.group) == int(force[4:])):
                raise RuntimeError(f'SHAPY_RCCL_FORCE_FAIL={force} (test hook)')
            self._comm = RcclComm.prepare(world=self.world if not dist.is_initialized() else None,
                                          group=self.group)
            if self._comm.world != self.world:
                raise RuntimeError(f'BetasGatherer(world={self.world}) on a group of {self._comm.world}')
        except Exception as e:                 # noqa: BLE001 -- any failure means "fall back"
            ok, err = 0, e
        ok = self._agree(ok)
        if ok:
            try:
                self._comm.connect()
            except Exception as e:             # noqa: BLE001
                ok, err = 0, e
            ok = self._agree(ok)
        if ok:
            return True
        if self._comm is not None:
            self._comm.close()
            self._comm = None
        logging.getLogger('shapy_amd.parallel').warning(
            'direct RCCL communicator unavailable (%s): falling back to torch.distributed\'s NCCL '
            'backend for the betas all-gather (one extra stream per process)', err)
        if not dist.is_initialized():
            raise RuntimeError('no RCCL communicator and no process group to fall back to') from err
        self.mode = 'work'
        if dist.get_backend(self.group) != 'nccl':
            self.group = self._fallback_group()                  # collective: every rank is here
        return False

    def _fallback_group(self):
        # only the members of self.group are here (the agreement ran over it): for a real subgroup the
        # creation must not wait for the other ranks of the world
        ranks = None if self.group is None else dist.get_process_group_ranks(self.group)
        if ranks is None or len(ranks) == dist.get_world_size():
            return dist.new_group(ranks=ranks, backend='nccl')
        return dist.new_group(ranks=ranks, backend='nccl', use_local_synchronization=True)

    def _call_c10d(self, local, out):
        work = dist.all_gather_into_tensor(out, local, group=self.group, async_op=True)
        self._pending = (out, work, local)
        return out

    def gather(self, local):
        out = self(local)
        self.wait()
        return out

    def close(self):
        """Joins the gather in flight and destroys the direct RCCL communicator (call before
        the process group goes away; idempotent)."""
        self.wait()
        if self._comm is not None:
            self._comm.close()
            self._comm = None


def gather_variable(local, group=None):
    """all_gather for shards of different length (last ranks may hold one item less)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = local.new_zeros((m,) + tuple(local.shape[1:]))
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
