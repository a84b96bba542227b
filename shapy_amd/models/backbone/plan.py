"""Plan of one HRNet forward: the flat op list over a liveness-packed activation workspace.

`HighResolutionNet._build_plan` (hrnet.py) walks the module tree and calls `_Plan.op` per kernel launch;
this module owns what follows from the op list alone: data dependencies between ops, the executor's
happens-before relation (lanes, barriers, dependency events, launch groups: csrc/hrnet_ops.hip), the event
slots the executor records / waits for, and the packing of the workspace under that relation.
"""
import numpy as np
import torch

class _Buf:
    __slots__ = ('H', 'W', 'C', 'size', 'uses', 'off', 'acc')

    def __init__(self, H, W, C):
        self.H, self.W, self.C = H, W, C
        self.size = H * W * C
        self.uses = []        # (epoch, lane, time index) of every op that touches the buffer
        self.acc = []         # (op index, is_write, first channel, end channel)
        self.off = None


class _Plan:
    """Flat op list + weight packing for one input resolution."""

    def __init__(self, bf16=False, x6=False):
        self.bf16 = bf16
        self.x6 = x6           # conv weights as three bf16 planes (utils/split.py), f32 otherwise
        self.ops = []          # dicts
        self.bufs = []
        self.wchunks = []      # byte strings, each padded to 16 bytes
        self.wbytes = 0
        self.epoch = 0
        self._pending_barrier = False
        self.cnt_ints = 0      # split-K arrival counters (int32 per image) handed out so far
        self._group_left = 0   # ops still to come in the current launch group
        self._group_t = 0      # ... and the group's time index (= op index of its first op)

    def padc(self, c):
        """bf16 rows are addressed in 16-byte slots of 8 channels (every HRNet width already is a
        multiple of 8; widths that are not a multiple of 32 -- the 48-channel branch -- take the
        flat-K kernel, csrc/conv_igemm.hip)."""
        return (c + 7) // 8 * 8 if self.bf16 else c

    def buf(self, H, W, C):
        b = _Buf(H, W, C)
        self.bufs.append(b)
        return b

    def barrier(self):
        self._pending_barrier = True

    def add_weights(self, arr, as_bf16=False):
        """Appends a tensor to the weight blob; returns its offset in ELEMENTS of its own type
        (float32, or bfloat16 for conv weights in bf16 mode)."""
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32).reshape(-1))
        if as_bf16:
            raw, esz = t.to(torch.bfloat16).view(torch.int16).numpy().tobytes(), 2
        else:
            raw, esz = t.numpy().tobytes(), 4
        off = self.wbytes // esz
        raw += b'\0' * ((-len(raw)) % 16)
        self.wchunks.append(raw)
        self.wbytes += len(raw)
        return off

    def add_conv_weights(self, w, x6=False):
        """OHWI conv weights in the layout of the plan's arithmetic (x6: of this layer's, in a float32 plan);
        offset in elements of the activation type (float32 words for f32 / f32x6, bfloat16 for bf16)."""
        if not (self.x6 or x6):
            return self.add_weights(w, as_bf16=self.bf16)
        from ...utils.split import split_bf16x3
        w = np.ascontiguousarray(w, np.float32)
        raw = split_bf16x3(w.reshape(w.shape[0], -1)).tobytes()      # [Cout, 3, Kp] bf16
        off = self.wbytes // 4
        raw += b'\0' * ((-len(raw)) % 16)
        self.wchunks.append(raw)
        self.wbytes += len(raw)
        return off

    def op(self, **kw):
        if self._pending_barrier:
            self.epoch += 1
            kw['barrier_before'] = 1
            self._pending_barrier = False
        else:
            kw.setdefault('barrier_before', 0)
        # the ops of a launch group run CONCURRENTLY (one persistent kernel): for the liveness
        # packing they all happen at the time of the group's first op
        kw.setdefault('group', 0)
        if self._group_left > 0:
            assert kw['group'] == 0 and not kw['barrier_before']
            self._group_left -= 1
            t = self._group_t
        else:
            t = len(self.ops)
            if kw['group'] > 1:
                self._group_left, self._group_t = kw['group'] - 1, t
        for key in ('inb', 'outb', 'resb', 'scrb'):
            b = kw.get(key)
            if b is not None:
                b.uses.append((self.epoch, kw['lane'], t))
        # data dependencies on earlier ops: read-after-write on the channels read (input: all,
        # residual: the output's slice), write-after-read / -write on the slice written
        i, deps = len(self.ops), set()
        acc = []
        if kw.get('inb') is not None:
            acc.append((kw['inb'], False, 0, kw['inb'].C))
        if kw.get('scrb') is not None:               # split-K slab: private scratch, written
            acc.append((kw['scrb'], True, 0, kw['scrb'].C))
        if kw.get('resb') is not None:
            acc.append((kw['resb'], False, kw['res_coff'], kw['res_coff'] + kw['Cout']))
        if kw.get('outb') is not None:
            acc.append((kw['outb'], True, kw['out_coff'], kw['out_coff'] + kw['Cout']))
        for b, is_w, c0, c1 in acc:
            for (j, w, a0, a1) in b.acc:
                if j != i and (is_w or w) and a0 < c1 and c0 < a1:
                    deps.add(j)
        for b, is_w, c0, c1 in acc:
            b.acc.append((i, is_w, c0, c1))
        kw['deps'] = deps
        self.ops.append(kw)

    def _walk(self):
        """Yields (first op index, member indices, mask implied by lane order and barriers) per
        launch: single ops and launch groups (whose members run concurrently), in plan order.
        Needs hb of earlier ops: used by happens_before / sync_plan while they fill it."""
        n, i = len(self.ops), 0
        lane_tail, barrier_mask = {}, 0
        while i < n:
            o = self.ops[i]
            g = max(1, o['group'])
            members = list(range(i, i + g))
            assert all(self.ops[k]['lane'] == o['lane'] for k in members)
            assert not any(self.ops[k]['barrier_before'] for k in members[1:])
            if o['barrier_before']:
                barrier_mask = (1 << i) - 1
            implied = barrier_mask | lane_tail.get(o['lane'], 0)
            yield i, members, implied
            tail = 0
            for k in members:
                tail |= self.hb[k] | (1 << k)
            lane_tail[o['lane']] = tail
            i += g

    def happens_before(self):
        """hb[i] = bit set of the ops that are guaranteed to have FINISHED when op i starts, under
        the executor's rules (csrc/hrnet_ops.hip): the ops of a lane run in plan order on the
        lane's stream (an op behind a launch group follows all its members); a barrier joins every
        lane (everything before it precedes everything after it); an op waits for the events of
        its data dependencies on other lanes; the members of a launch group run concurrently."""
        self.hb = [0] * len(self.ops)
        for _, members, implied in self._walk():
            for k in members:
                m = implied
                for d in self.ops[k]['deps']:
                    assert d not in members, 'members of a launch group must be independent'
                    m |= self.hb[d] | (1 << d)
                self.hb[k] = m
        return self.hb

    def sync_plan(self, max_events=64):
        """Event slots for the executor: per op `sig` (slot to record after it, or -1) and `wait`
        (<= 3 slots to wait for before it) = the op's dependencies on OTHER lanes that are not
        already implied by its lane order, a barrier or another of its dependencies."""
        n = len(self.ops)
        self.hb = [0] * n
        waits = [[] for _ in range(n)]
        for _, members, implied in self._walk():
            for k in members:
                o = self.ops[k]
                m = implied
                for d in o['deps']:
                    m |= self.hb[d] | (1 << d)
                self.hb[k] = m
                cross = sorted(d for d in o['deps']
                               if self.ops[d]['lane'] != o['lane'] and not (implied >> d) & 1)
                need = [d for d in cross if not any(d2 != d and (self.hb[d2] >> d) & 1 for d2 in cross)]
                if len(need) > 3:
                    raise ValueError(f"op {k} ({o.get('name')}) waits for {len(need)} lanes")
                waits[k] = need
        # slots: an event is free again once its last waiter has been enqueued (plan order)
        last_waiter = {}
        for i, w in enumerate(waits):
            for d in w:
                last_waiter[d] = i
        free, busy, sig = list(range(max_events)), [], [-1] * n
        for i in range(n):
            busy.sort()
            while busy and busy[0][0] < i:
                free.append(busy.pop(0)[1])
            if i in last_waiter:
                if not free:
                    raise ValueError('out of dependency events')
                sig[i] = free.pop(0)
                busy.append((last_waiter[i], sig[i]))
        for i, o in enumerate(self.ops):
            o['sig'] = sig[i]
            o['wait'] = [sig[d] for d in waits[i]] + [-1] * (3 - len(waits[i]))
        return waits

    def allocate(self):
        """Packing of the activation workspace (floats per image) under the plan's happens-before
        order: two buffers may share memory iff EVERY op that touches one has finished before ANY op
        that touches the other starts (lane order, barriers, dependency events: `happens_before`).
        First fit in order of first use."""
        hb = self.happens_before()
        used = [b for b in self.bufs if b.acc]
        touch = {id(b): sorted({j for (j, _, _, _) in b.acc}) for b in used}
        mask = {id(b): sum(1 << j for j in touch[id(b)]) for b in used}

        # common[b] = the ops that precede EVERY use of b
        common = {}
        for b in used:
            c = -1
            for v in touch[id(b)]:
                c &= hb[v]
            common[id(b)] = c

        def before(a, b):          # every use of a precedes every use of b
            ma = mask[id(a)]
            return (common[id(b)] & ma) == ma
        placed, total = [], 0
        for b in sorted(used, key=lambda b: (touch[id(b)][0], -b.size)):
            conflicts = sorted((p.off, p.size) for p in placed if not (before(p, b) or before(b, p)))
            off = 0
            for (o, sz) in conflicts:
                if off + b.size <= o:
                    break
                off = max(off, (o + sz + 7) // 8 * 8)      # aligned BEFORE the next gap test
            b.off = off
            placed.append(b)
            total = max(total, off + b.size)
        return (total + 7) // 8 * 8
