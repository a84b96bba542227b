"""Per-shape microbenchmark of the MFMA conv kernel over the conv classes of HRNet-W48.

    python tools/conv_bench.py [--batch 64] [--size 224] [--tiles auto,wino,wino4,256x48,...] [--out file]

For every distinct (Hi, Cin, Cout, ksize, stride, ups, residual) class of the backbone's op
list it times shapy_conv2d alone on the GPU (HIP events, single stream) and prints
count x time, TFLOP/s and the share of the summed time -- the tuning table behind
conv_tile_auto() in csrc/conv_igemm.hip.
"""
import argparse
import collections
import ctypes
import json
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402

from shapy_amd import _lib  # noqa: E402
from shapy_amd.config import default_config  # noqa: E402
from shapy_amd.models.backbone.hrnet import HighResolutionNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--tiles', default='auto')
    ap.add_argument('--iters', type=int, default=8)
    ap.add_argument('--out', default='')
    ap.add_argument('--filter', default='', help='Hi,Cin,Cout,ksize: run only this class')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'f32x6'])
    ap.add_argument('--wino4-min-hw', type=int, default=7,
                    help='tile "wino4": only maps of at least this many pixels a side')
    args = ap.parse_args()
    lib = _lib.load()
    net = HighResolutionNet(default_config().network.smplx.backbone.hrnet)
    bf16 = args.dtype == 'bf16'
    tdt = torch.bfloat16 if bf16 else torch.float32
    P = net._build_plan(args.size, args.size, bf16)
    classes = collections.OrderedDict()
    for o in P.ops:
        if o['type'] != _lib.OP_CONV:
            continue
        key = (o['Hi'], o['Wi'], o['Cin'], o['Cout'], o['ksize'], o['stride'], o['ups'],
               o['resb'] is not None, o['relu'])
        classes[key] = classes.get(key, 0) + 1
    B = args.batch
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = []
    flt = [int(v) for v in args.filter.split(',')] if args.filter else None
    for key, count in classes.items():
        Hi, Wi, Cin, Cout, ks, st, ups, has_res, relu = key
        if flt and [Hi, Cin, Cout, ks] != flt:
            continue
        pad = ks // 2
        Ho, Wo = (Hi + 2 * pad - ks) // st + 1, (Wi + 2 * pad - ks) // st + 1
        x = torch.randn(B, Hi, Wi, Cin, device='cuda').to(tdt)
        w = (torch.randn(Cout, ks, ks, Cin, device='cuda') * 0.05).to(tdt)
        if args.dtype == 'f32x6':                      # three bf16 planes [Cout, 3, Kp]
            from shapy_amd.utils.split import split_bf16x3
            import numpy as np
            w = torch.from_numpy(split_bf16x3(w.cpu().numpy().reshape(Cout, -1)).view(np.int16)).cuda()
        b = torch.randn(Cout, device='cuda')
        out = torch.empty(B, Ho * ups, Wo * ups, Cout, device='cuda', dtype=tdt)
        res = torch.randn_like(out) if has_res else None
        flop = 2.0 * B * Ho * Wo * Cout * Cin * ks * ks
        wu = wu4 = None
        for tile in args.tiles.split(','):
            d = _lib.ShapyConv()
            if tile.startswith('autok'):               # implicit GEMM, automatic tile, split-K with S slices
                pass
            elif tile == 'wino4' or tile.startswith('wino4k'):   # Winograd F(4x4,3x3) (conv_wino4.hip); wino4kS: split-K, S slices
                from shapy_amd.utils import winograd
                if args.dtype != 'f32' or not winograd.eligible4(ks, st, pad, Cin, Cout, ups) \
                        or min(Hi, Wi) < args.wino4_min_hw:
                    continue
                if wu4 is None:
                    wu4 = torch.from_numpy(winograd.transform_filters4(w.cpu().numpy())).cuda()
                d.wgt_wino = wu4.data_ptr()
            elif tile in ('wino', 'wino1', 'wino2', 'winochunk'):   # Winograd F(2x2,3x3) where it applies
                from shapy_amd.utils import winograd
                if args.dtype != 'f32' or not winograd.eligible(ks, st, pad, Cin, Cout, ups):
                    continue
                if wu is None:
                    wu = torch.from_numpy(winograd.transform_filters(w.cpu().numpy())).cuda()
                d.wgt_wino = wu.data_ptr()
            d.in_ = x.data_ptr(); d.wgt = w.data_ptr(); d.bias = b.data_ptr()
            d.res = res.data_ptr() if has_res else None
            d.out = out.data_ptr()
            d.B, d.Hi, d.Wi, d.Cin, d.in_ld = B, Hi, Wi, Cin, Cin
            d.Ho, d.Wo, d.Cout = Ho, Wo, Cout
            d.ksize, d.stride, d.pad = ks, st, pad
            d.out_ld = Cout; d.out_coff = 0; d.res_ld = Cout if has_res else 0; d.res_coff = 0
            d.relu = int(relu); d.ups = ups
            if tile.startswith('autok'):
                S = int(tile[5:])
                if ups != 1 or (bf16 and Cin % 32):
                    continue
                slab, ncnt = _lib.igemm_split_sizes(Ho, Wo, Cout, S)
                split_ws = torch.empty(slab * B, device='cuda')
                split_cnt = torch.zeros(ncnt * B, dtype=torch.int32, device='cuda')
                d.split_ws, d.split_cnt = split_ws.data_ptr(), split_cnt.data_ptr()
                d.split_kib, d.split_cnt_n = split_ws.numel() * 4 // 1024, split_cnt.numel()
                d.tile = 0x2000 | _lib.tile_w4_ksplit(S)
            elif tile.startswith('wino4k'):
                S = int(tile[6:])
                if (Cin // 16) % S:
                    continue
                slab, ncnt = _lib.w4_split_sizes(Hi, Wi, Cout, S)
                split_ws = torch.empty(slab * B, device='cuda')
                split_cnt = torch.zeros(ncnt * B, dtype=torch.int32, device='cuda')
                d.split_ws, d.split_cnt = split_ws.data_ptr(), split_cnt.data_ptr()
                d.split_kib, d.split_cnt_n = split_ws.numel() * 4 // 1024, split_cnt.numel()
                d.tile = _lib.TILE_WINO4 | _lib.tile_w4_ksplit(S)
            else:
                d.tile = {'wino': 0, 'wino1': 0x4000, 'wino2': 0x8000, 'winochunk': 0x20000,
                          'wino4': _lib.TILE_WINO4}[tile] if tile.startswith('wino') else _lib.TILES[tile]
            d.dtype = {'f32': _lib.DTYPE_F32, 'bf16': _lib.DTYPE_BF16, 'f32x6': _lib.DTYPE_F32X6}[args.dtype]
            rc = 0
            for _ in range(2):
                rc = lib.shapy_conv2d(ctypes.byref(d), stream)
            if rc == -1:                               # tile not available for this layer shape
                continue
            assert rc == 0, (rc, key, tile)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                lib.shapy_conv2d(ctypes.byref(d), stream)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            rows.append(dict(key=key, count=count, tile=tile, us=us, tflops=flop / us / 1e6,
                             gflop=flop / 1e9, M=B * Ho * Wo))
    best = {}
    for r in rows:
        if r['key'] not in best or r['us'] < best[r['key']]['us']:
            best[r['key']] = r
    tot = sum(r['us'] * r['count'] for r in best.values())
    totf = sum(r['gflop'] * r['count'] for r in best.values())
    print(f'# B={B} size={args.size}: sum over classes of best time = {tot / 1e3:.2f} ms, '
          f'{totf / tot / 1e3:.1f} TFLOP/s')
    print('# Hi Cin->Cout k s ups res | count | M | tile: us (TFLOP/s) ... | share of total')
    for key in classes:
        rs = [r for r in rows if r['key'] == key]
        if not rs:
            continue
        Hi, Wi, Cin, Cout, ks, st, ups, has_res, relu = key
        s = ' '.join(f"{r['tile']}:{r['us']:.0f}us({r['tflops']:.0f})" for r in rs)
        bb = best[key]
        print(f'{Hi:3d} {Cin:4d}->{Cout:4d} k{ks} s{st} u{ups} r{int(has_res)} | x{classes[key]:3d} | '
              f'M={bb["M"]:6d} | {s} | {100 * bb["us"] * bb["count"] / tot:.1f}%')
    if args.out:
        with open(args.out, 'w') as f:
            json.dump([dict(r, key=list(r['key'])) for r in rows], f)


if __name__ == '__main__':
    main()
