"""Software pipelining of consecutive batches: the NEXT batch's stem + layer1 under the CURRENT batch.

`HighResolutionNet.forward(x, prefetch=next_x)` cuts the op list at its first barrier (transition1): the ops in
front of it -- stem, conv2, the four Bottlenecks of layer1 -- are one HBM-bound chain on one lane, while stage 4
and the head of the running batch are matrix-core work that leaves the memory system idle.  The prologue of
`next_x` is issued on one of the executor's own side streams (no new stream: HIP multiplexes streams onto four
hardware queues and a fifth one serialises the lanes) into a SECOND activation workspace;
the next `forward(next_x)` finds it (same tensor, same version counter, same plan, same caller stream), waits
for its event and runs only the rest of the op list.  Anything else -- another tensor, an in-place edit, a
rebuilt plan -- is a full forward in the other workspace and the stale prologue is never looked at.

Same kernels, same order per image: features are bit-identical with and without (tests/test_gpu_parity.py).
Measured (profiles/r05o_prologue_prefetch_ab.txt, backbone only): float32 B = 64 12.45 -> 12.17 ms, B = 32 7.15 ->
6.86, B = 16 4.96 -> 4.68, B = 8 3.86 -> 3.66; bf16 B = 32 3.72 -> 3.50, B = 64 6.11 -> 5.92.  Up to B = 16 the
prologue is issued BEFORE the rest (a small batch leaves lane 3 idle through stages 2-3; behind lane 3's own
stage-4 work it would wait for the whole latency chain: B = 8 4.18 ms), larger batches issue it behind the rest.
Small batches are chains of latency-bound launches, and a second chain beside the first is nearly free: up to
B = 16 the cut is the SECOND barrier (stem + layer1 + stage 2 as the prologue, 35 ops): B = 1 3.21 -> 2.80 ms,
B = 8 3.83 -> 3.39, B = 16 5.00 -> 4.36; at B = 64 that cut loses (12.32 against 12.17), and everything in front of
stage 4 (160 serial ops) loses at every size.
"""
import ctypes

import torch

from ... import _lib

LANE = 3                 # the 4th branch's stream: idle until stage 4


def cut_of(plan, nth=1):
    """Index of the first op of the 'rest' -- the plan's nth barrier (1: transition1, the prologue is stem + layer1;
    2: transition2, + stage 2) --, or 0 when the ops in front of it cannot run as a self-contained serial prologue:
    a later op waits for one of their events (the prologue records none).  Split-K layers may be among them: every
    workspace has arrival counters of its own."""
    ops = plan.ops
    bars = [i for i, o in enumerate(ops) if o['barrier_before']]
    if len(bars) < nth:
        return 0
    cut = bars[nth - 1]
    sigs = {o['sig'] for o in ops[:cut] if o['sig'] >= 0}
    for o in ops[cut:]:                      # (event slots are reused: a slot signalled again belongs to the rest)
        if any(w in sigs for w in o['wait']):
            return 0
        sigs.discard(o['sig'])
        if not sigs:
            break
    return cut


def ops_from(eng, first):
    if not first:
        return eng['ops']
    return ctypes.cast(ctypes.byref(eng['ops'], first * ctypes.sizeof(_lib.ShapyOp)),
                       ctypes.POINTER(_lib.ShapyOp))


class ProloguePrefetch:
    """The stash of one network: at most one prologue in flight."""

    def __init__(self):
        self.pending = None
        self._side = {}
        self.issued = self.used = 0

    def side_stream(self, lib, device):
        s = self._side.get(device.index)
        if s is None:
            h = ctypes.c_void_p()
            _lib.check(lib.shapy_hrnet_lane_stream(LANE, ctypes.byref(h)), 'shapy_hrnet_lane_stream')
            s = self._side[device.index] = torch.cuda.ExternalStream(h.value, device=device)
        return s

    @staticmethod
    def key(x, sk):
        # (an inference tensor has no version counter -- reading it raises; such tensors are never stashed, `usable`)
        return (x.data_ptr(), tuple(x.shape), None if x.is_inference() else x._version, sk)

    def drop(self):
        """Forget the stashed prologue (the next forward runs in full).  For callers that WRITE the handed-over
        tensor behind torch's back between the two calls: the stash is recognised by (address, shape, torch
        version counter), and a raw-pointer kernel, a DLPack alias or another process's copy engine changes the
        bytes without bumping the counter (HighResolutionNet.drop_prefetch, tests/test_gpu_parity.py::
        test_hrnet_prefetch_stash_rules)."""
        self.pending = None

    def take(self, x, eng, ent, sk):
        """The stashed prologue of exactly this input, or None (the stash is dropped either way).  The stash holds
        the tensor, the plan and the workspace entry themselves: none of their addresses can have been reused.
        A stash that does NOT match may still be running: the caller's stream waits for it before the forward that
        follows may hand its workspace to the next prologue."""
        pf, self.pending = self.pending, None
        if pf is not None and pf['eng'] is eng and pf['ent'] is ent and pf['key'] == self.key(x, sk):
            self.used += 1
            return pf
        if pf is not None and pf['sk'] == sk:
            torch.cuda.current_stream().wait_event(pf['done'])
        return None

    @staticmethod
    def usable(nx, x):
        return (torch.is_tensor(nx) and nx.is_cuda and nx.device == x.device and nx.dtype == torch.float32
                and nx.shape == x.shape and nx.is_contiguous() and not nx.is_inference())

    @staticmethod
    def second_workspace(ent, need, device):
        """The other workspace and its split-K arrival counters, allocated (and the counters ZEROED, on the caller's
        stream) before `ready_event` is recorded: the side stream waits for that event only, so a memset enqueued
        behind it could still be pending when the first prologue reaches a split-K layer."""
        other = 1 - ent['cur']
        if ent['ws'][other] is None:
            ent['ws'][other] = torch.empty(max(need, ent['ws'][0].numel()), dtype=torch.uint8, device=device)
            c0 = ent['cnt'][0]                     # the second workspace has arrival counters of its own
            ent['cnt'][other] = None if c0 is None else torch.zeros_like(c0)
            ent['done'] = [torch.cuda.Event(), torch.cuda.Event()]

    def ready_event(self, ent):
        """Recorded on the caller's stream BEFORE the rest of the running batch is issued: the next input is
        there, the other workspace's last user (the batch before this one) is done and its counters are zero."""
        if 'ev' not in ent:
            ent['ev'] = torch.cuda.Event()
        ent['ev'].record()
        return ent['ev']

    def issue(self, lib, run, eng, ent, nx, ev, sk, need, cut):
        other = 1 - ent['cur']
        assert ent['ws'][other] is not None, 'second_workspace() runs before ready_event()'
        side = self.side_stream(lib, nx.device)
        side.wait_event(ev)
        # the side stream reads / writes these after the caller may have dropped them
        nx.record_stream(side)
        ent['ws'][other].record_stream(side)
        if ent['cnt'][other] is not None:
            ent['cnt'][other].record_stream(side)
        rc = run(0, cut, nx, other, 0, ctypes.c_void_p(side.cuda_stream))
        if rc != 0:
            return rc
        ent['done'][other].record(side)
        self.pending = dict(key=self.key(nx, sk), eng=eng, ent=ent, sk=sk, arena=other, done=ent['done'][other], x=nx,
                            cut=cut)
        self.issued += 1
        return 0
