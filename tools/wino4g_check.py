"""Canary + parity + timing of the persistent grouped F(4x4) launch (csrc/conv_wino4g.hip) against
one shapy_conv2d (conv_wino4.hip) per layer: same tasks, same products; NaN pattern (untouched channels of a
concat epilogue) identical, values equal to float32 rounding (since round 6 the per-layer kernel adds the Winograd
rows 4 / 5 of a tile in another association: its fourth wave's partial x-transform).

    timeout 120 python tools/wino4g_check.py --canary      # tiny cases first (a protocol bug hangs)
    python tools/wino4g_check.py --bench                   # HRNet level groups at B = 64
"""
import argparse
import ctypes
import os.path as osp
import sys
import time

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from shapy_amd import _lib  # noqa: E402
from shapy_amd.utils import winograd  # noqa: E402




def make_desc(B, H, W, C, O, res, relu, coff=0, extra_ld=0, g=None):
    x = torch.randn(B, H, W, C, generator=g).cuda()
    w = (torch.randn(O, 3, 3, C, generator=g) / np.sqrt(9 * C)).cuda()
    b = torch.randn(O, generator=g).cuda()
    wu = torch.from_numpy(winograd.transform_filters4(w.cpu().numpy())).cuda()
    ld = O + extra_ld
    r = torch.randn(B, H, W, ld, generator=g).cuda() if res else None
    out = torch.full((B, H, W, ld), float('nan'), device='cuda')
    d = _lib.ShapyConv()
    d.dtype = _lib.DTYPE_F32
    d.in_ = x.data_ptr(); d.wgt = w.data_ptr(); d.bias = b.data_ptr()
    d.res = r.data_ptr() if res else None
    d.out = out.data_ptr()
    d.B, d.Hi, d.Wi, d.Cin, d.in_ld = B, H, W, C, C
    d.Ho, d.Wo, d.Cout = H, W, O
    d.ksize, d.stride, d.pad = 3, 1, 1
    d.out_ld = ld; d.out_coff = coff; d.res_ld = ld if res else 0; d.res_coff = coff if res else 0
    d.relu = int(relu); d.ups = 1; d.tile = _lib.TILE_WINO4
    d.wgt_wino = wu.data_ptr()
    return d, dict(x=x, w=w, b=b, wu=wu, r=r, out=out)


def run_case(lib, shapes, seed, stream):
    g = torch.Generator().manual_seed(seed)
    descs, keep = [], []
    for s in shapes:
        d, k = make_desc(*s, g=g)
        descs.append(d); keep.append(k)
    # reference: one launch per layer
    refs = []
    for d, k in zip(descs, keep):
        k['out'].fill_(float('nan'))
        rc = lib.shapy_conv2d(ctypes.byref(d), stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
        refs.append(k['out'].clone())
        k['out'].fill_(float('nan'))
    arr = (_lib.ShapyConv * len(descs))(*descs)
    rc = lib.shapy_conv2d_group(arr, len(descs), stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    ok = True
    for i, (k, ref) in enumerate(zip(keep, refs)):
        got = k['out']
        g0, r0 = torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(ref, nan=-7.0)
        tol = 2e-5 * max(1.0, float(r0.abs().max()))
        same = bool(torch.equal(torch.isnan(got), torch.isnan(ref)) and float((g0 - r0).abs().max()) <= tol)
        ok &= same
        if not same:
            bad = (g0 - r0).abs() > tol
            print('   MISMATCH conv', i, shapes[i], 'elements', int(bad.sum()), 'of', bad.numel(),
                  'first', bad.nonzero()[:3].tolist())
    print(('ok  ' if ok else 'FAIL'), shapes)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--canary', action='store_true')
    ap.add_argument('--bench', action='store_true')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ok = True
    if args.canary:
        cases = [
            [(1, 8, 8, 16, 48, False, True)],                                  # 1 task
            [(2, 12, 20, 48, 48, True, True)],                                 # 2 tasks, 3 chunks
            [(1, 7, 9, 32, 96, True, False)],                                  # generic loop, 2 n tiles
            [(1, 8, 8, 16, 48, False, True), (1, 4, 4, 32, 48, True, True)],   # two convs
            [(2, 28, 28, 96, 96, True, True), (2, 56, 56, 48, 48, True, True),
             (2, 14, 14, 192, 192, True, True), (2, 7, 7, 384, 384, True, True)],
            [(1, 14, 14, 16, 48, False, False, 16, 32)],                       # channel offset epilogue
            [(3, 20, 12, 96, 144, True, True), (5, 9, 9, 48, 96, False, True)],   # nbx = 3: generic split
            [(40, 56, 56, 48, 48, True, True)],                                # more tasks than workgroups
        ]
        for i, c in enumerate(cases):
            ok &= run_case(lib, c, 10 + i, stream)
        # the same group again and again
        for rep in range(3):
            ok &= run_case(lib, cases[4], 50 + rep, stream)
        print('CANARY', 'OK' if ok else 'FAILED')
    if args.bench:
        B = args.batch
        levels = {
            'stage2 (48,96)': [(B, 56, 56, 48, 48, True, True), (B, 28, 28, 96, 96, True, True)],
            'stage3 (48,96,192)': [(B, 56, 56, 48, 48, True, True), (B, 28, 28, 96, 96, True, True),
                                   (B, 14, 14, 192, 192, True, True)],
            'stage4 (48,96,192,384)': [(B, 56, 56, 48, 48, True, True), (B, 28, 28, 96, 96, True, True),
                                       (B, 14, 14, 192, 192, True, True), (B, 7, 7, 384, 384, True, True)],
            '48 alone': [(B, 56, 56, 48, 48, True, True)],
            '96 alone': [(B, 28, 28, 96, 96, True, True)],
            '192 alone': [(B, 14, 14, 192, 192, True, True)],
            '384 alone': [(B, 7, 7, 384, 384, True, True)],
            '256->48': [(B, 56, 56, 256, 48, False, True)],
        }
        for name, shapes in levels.items():
            g = torch.Generator().manual_seed(1)
            descs, keep = [], []
            for s in shapes:
                d, k = make_desc(*s, g=g)
                descs.append(d); keep.append(k)
            arr = (_lib.ShapyConv * len(descs))(*descs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def single():
                for d in descs:
                    assert lib.shapy_conv2d(ctypes.byref(d), stream) == 0

            def group():
                assert lib.shapy_conv2d_group(arr, len(descs), stream) == 0
            res = {}
            for nm, fn in (('per-layer', single), ('grouped', group)):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[nm] = e0.elapsed_time(e1) / args.iters * 1e3
            exec_flop = sum(2 * 36 * s[0] * -(-s[1] // 4) * -(-s[2] // 4) * s[3] * s[4] for s in shapes)
            print(f'{name:26s} per-layer (one stream) {res["per-layer"]:7.1f} us   grouped {res["grouped"]:7.1f} us'
                  f'   executed MFMA {exec_flop / res["grouped"] / 1e6:6.1f} TF/s '
                  f'({exec_flop / res["grouped"] / 1e6 / 157.3:.2f} of peak)')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
