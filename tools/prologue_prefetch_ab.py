"""A/B of running the NEXT batch's stem + layer1 under the CURRENT batch's stages / head (backbone only).

    python tools/prologue_prefetch_ab.py [--batch 64] [--steps 30] [--dtype f32]

The op list is cut at the first barrier (transition1: ops before it are the stem and layer1, HBM-bound, one
lane); the prologue of step k + 1 runs in a second workspace on one of the executor's side streams while the
rest of step k runs as usual.  Modes:
  base      the product forward, one at a time
  product   the product's own pipelined loop: net(x, prefetch=next_x) (shapy_amd/models/backbone/prefetch.py)
  split     prologue and rest as two calls on the caller's stream (cost of the cut itself)
  before-L  prologue of k + 1 enqueued on lane L's stream BEFORE the rest of k is issued (overlaps stages 2-3)
  after-L   ... AFTER the rest of k (stream order puts it behind lane L's own work: overlaps stage 4 / head)
Features of every mode are compared with the product forward (same kernels, same order per image: bit-equal).
"""
import argparse
import ctypes
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402

from shapy_amd import _lib  # noqa: E402
from shapy_amd.config import default_config  # noqa: E402
from shapy_amd.models.backbone.hrnet import HighResolutionNet  # noqa: E402
from shapy_amd.utils import synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--barrier', type=int, default=1, help='cut at the N-th barrier of the plan (1: transition1, 2: transition2)')
    ap.add_argument('--modes', default='base,product,split,before-3,after-3,after-1,before-2,base')
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda:0')
    net = HighResolutionNet(default_config().network.smplx.backbone.hrnet).to(dev).eval()
    net.compute_dtype = args.dtype
    B, S = args.batch, args.size
    xs = [torch.from_numpy(syn.synthetic_images(B, S, 11 + i)).to(dev) for i in range(2)]
    ref = [net(x)['concat'].clone() for x in xs]
    eng = net._compile(S, S, dev, graph=False, B=B)
    P = eng['plan']
    cut = [i for i, o in enumerate(P.ops) if o['barrier_before']][args.barrier - 1]
    live = {o['sig'] for o in P.ops[:cut] if o['sig'] >= 0}
    for o in P.ops[cut:]:                    # (event slots are reused: a slot signalled again belongs to the rest)
        assert not any(w in live for w in o['wait']), 'the rest waits for a prologue event'
        live.discard(o['sig'])
    # (split-K layers in the prologue are fine HERE: prologue and rest are disjoint op sets, i.e. disjoint counter slices)
    print(f'# cut at op {cut} ({P.ops[cut].get("name")}): prologue {cut} ops, rest {eng["n_ops"] - cut}')
    need = eng['ws_per_img'] * B * eng['esz']
    arenas = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in range(2)]
    cnt = net._counters(eng, B, dev)
    ops_rest = ctypes.cast(ctypes.byref(eng['ops'], cut * ctypes.sizeof(_lib.ShapyOp)), ctypes.POINTER(_lib.ShapyOp))
    main_s = torch.cuda.current_stream()

    def lane(L):
        h = ctypes.c_void_p()
        _lib.check(lib.shapy_hrnet_lane_stream(L, ctypes.byref(h)), 'lane_stream')
        return torch.cuda.ExternalStream(h.value, device=dev)

    def run(ops, n, x, ws, feat, multi, stream):
        rc = lib.shapy_hrnet_run(ops, n, _lib.ptr(eng['weights']), _lib.ptr(x), _lib.ptr(ws), eng['ws_per_img'],
                                 _lib.ptr(cnt), eng['cnt_per_img'], _lib.ptr(feat), B, S, S, multi, eng['dtype'],
                                 ctypes.c_void_p(stream.cuda_stream))
        _lib.check(rc, 'shapy_hrnet_run')

    def loop(mode, steps):
        feats = [torch.empty(B, eng['feat_dim'], device=dev) for _ in range(2)]
        if mode == 'base':
            for k in range(steps):
                feats[k & 1] = net(xs[k & 1])['concat']
            return feats
        if mode == 'product':
            for k in range(steps):
                feats[k & 1] = net(xs[k & 1], prefetch=xs[(k + 1) & 1])['concat']
            return feats
        if mode == 'split':
            for k in range(steps):
                run(eng['ops'], cut, xs[k & 1], arenas[0], feats[k & 1], 0, main_s)
                run(ops_rest, eng['n_ops'] - cut, xs[k & 1], arenas[0], feats[k & 1], 1, main_s)
            return feats
        when, L = mode.split('-')
        side = lane(int(L))
        done = [torch.cuda.Event(), torch.cuda.Event()]

        def ready():                          # on the caller's stream BEFORE the rest of the running step is
            ev = torch.cuda.Event()           # issued: the next input is there and the arena's previous user
            ev.record(main_s)                 # (step k - 2) is done
            return ev

        def prologue(k, ev):                  # step k's stem + layer1 into arena k & 1, on the side stream
            side.wait_event(ev)
            run(eng['ops'], cut, xs[k & 1], arenas[k & 1], feats[k & 1], 0, side)
            done[k & 1].record(side)
        prologue(0, ready())
        for k in range(steps):
            ev = ready()
            if when == 'before' and k + 1 < steps:
                prologue(k + 1, ev)
            main_s.wait_event(done[k & 1])
            run(ops_rest, eng['n_ops'] - cut, xs[k & 1], arenas[k & 1], feats[k & 1], 1, main_s)
            if when == 'after' and k + 1 < steps:
                prologue(k + 1, ev)
        return feats

    for mode in args.modes.split(','):
        with torch.no_grad():
            loop(mode, 4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            feats = loop(mode, args.steps)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        # steps is even: feats[1] is the last step's output (input 1), feats[0] the one before
        err = max(float((feats[i] - ref[i]).abs().max()) for i in range(2))
        print(f'{mode:10s} {ms:8.3f} ms/step  {B / ms * 1e3:8.1f} img/s   max |features - product| = {err:.2e}',
              flush=True)


if __name__ == '__main__':
    main()
