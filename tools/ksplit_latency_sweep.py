"""Backbone latency per batch size under F(4x4) split-K policies (HighResolutionNet.wino4_ksplit), one process:
the network is built once, every policy compiles its own plan, every (policy, batch) pair is timed with HIP events
over `--iters` forwards after a warm-up.

    python tools/ksplit_latency_sweep.py [--batches 1,4,8,16,32,64] [--iters 30]
"""
import argparse
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402
from shapy_amd.utils import synthetic as syn  # noqa: E402

POLICIES = [
    ('none', {}),
    ('384@4:2', {(384, 4): 2}),
    ('384@4:4', {(384, 4): 4}),
    ('384@4:4,192@16:2', {(384, 4): 4, (192, 16): 2}),
    ('384@4:4,192@16:4', {(384, 4): 4, (192, 16): 4}),
    ('384@4:4,192@16:4,96@49:2', {(384, 4): 4, (192, 16): 4, (96, 49): 2}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='1,4,8,16,32,64')
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--size', type=int, default=224)
    args = ap.parse_args()
    net, _ = ge.make_network()
    bb = net.backbone
    batches = [int(b) for b in args.batches.split(',')]
    xs = {b: torch.from_numpy(syn.synthetic_images(b, args.size, 100)).cuda() for b in batches}
    print('policy'.ljust(28) + ''.join(f'B={b}'.rjust(9) for b in batches) + '   (backbone ms)')
    ref = {}
    for name, pol in POLICIES:
        bb.wino4_ksplit = dict(pol)
        row = []
        for b in batches:
            x = xs[b]
            with torch.no_grad():
                for _ in range(5):
                    f = bb(x)['concat']
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    f = bb(x)['concat']
                e1.record()
                torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / args.iters)
            if name == 'none':
                ref[b] = f.clone()
            else:
                err = (f - ref[b]).abs().max().item() / max(1.0, ref[b].abs().max().item())
                assert err < 2e-5, (name, b, err)
        print(name.ljust(28) + ''.join(f'{t:9.3f}' for t in row))


if __name__ == '__main__':
    main()
