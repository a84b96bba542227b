"""Backbone latency per batch size under split-K policies of the IMPLICIT-GEMM layers (HighResolutionNet.direct_ksplit),
float32 (the F(4x4) layers keep their batch-bucketed policy) or bf16 storage; one process, one plan per policy.

    python tools/direct_ksplit_sweep.py --dtype bf16 --batches 32,64
    python tools/direct_ksplit_sweep.py --dtype f32 --batches 1,8,32,64
"""
import argparse
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402
from shapy_amd.utils import synthetic as syn  # noqa: E402

# keys: (Cin, ksize, most output pixels per image)
BF16 = [
    ('none', {}),
    ('384@7x7:2', {(384, 3, 49): 2}),
    ('384@7x7:4', {(384, 3, 49): 4}),
    ('384:4,192@14x14:2', {(384, 3, 49): 4, (192, 3, 196): 2}),
    ('384:4,192:2,head', {(384, 3, 49): 4, (192, 3, 196): 2, (2048, 1, 49): 2, (1536, 1, 49): 2, (512, 3, 49): 4}),
    ('384:4,192:4,head,96@28:2', {(384, 3, 49): 4, (192, 3, 196): 4, (2048, 1, 49): 2, (1536, 1, 49): 2,
                                  (512, 3, 49): 4, (96, 3, 784): 2}),
]
F32 = [
    ('none', {}),
    ('head 1x1:2', {(2048, 1, 49): 2, (1536, 1, 49): 2}),
    ('head 1x1:4,512:2', {(2048, 1, 49): 4, (1536, 1, 49): 4, (512, 1, 49): 2}),
    ('head:4,s2 192:2', {(2048, 1, 49): 4, (1536, 1, 49): 4, (512, 1, 49): 2, (192, 3, 49): 2}),
    ('head:4,s2 192:4,96:2', {(2048, 1, 49): 4, (1536, 1, 49): 4, (512, 1, 49): 2, (192, 3, 49): 4, (96, 3, 196): 2,
                              (96, 3, 49): 2}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16', choices=['f32', 'bf16'])
    ap.add_argument('--batches', default='32,64')
    ap.add_argument('--iters', type=int, default=30)
    args = ap.parse_args()
    net, _ = ge.make_network()
    bb = net.backbone
    bb.compute_dtype = args.dtype
    batches = [int(b) for b in args.batches.split(',')]
    xs = {b: torch.from_numpy(syn.synthetic_images(b, 224, 100)).cuda() for b in batches}
    print(f'{args.dtype}: policy'.ljust(34) + ''.join(f'B={b}'.rjust(9) for b in batches) + '   (backbone ms)')
    ref = {}
    for name, pol in (BF16 if args.dtype == 'bf16' else F32):
        bb.direct_ksplit = dict(pol)
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
                assert err < (3e-2 if args.dtype == 'bf16' else 2e-5), (name, b, err)
        print(name.ljust(34) + ''.join(f'{t:9.3f}' for t in row))


if __name__ == '__main__':
    main()
