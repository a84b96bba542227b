"""Tile sweep of the SMPL-X pose-blend GEMM shape on shapy_conv2d: out[B, N] = in[B, K] . W[N, K]^T (+ residual),
K = Ppad = 512, N = 31,425 (V * 3), B = 4 .. 64 -- the skinny-M GEMM that streams posedirs (61 MB) once per batch.

    python tools/skinny_gemm_bench.py [--batch 64]
"""
import argparse
import ctypes
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402

from shapy_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--K', type=int, default=512)
    ap.add_argument('--N', type=int, default=31425)
    ap.add_argument('--iters', type=int, default=50)
    args = ap.parse_args()
    lib = _lib.load()
    B, K, N = args.batch, args.K, args.N
    x = torch.randn(B, K, device='cuda')
    w = torch.randn((N + 127) // 128 * 128, K, device='cuda') * 0.05
    res = torch.randn(B, N, device='cuda')
    out = torch.empty(B, N, device='cuda')
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = None
    for tile in ('auto', 'auto+pd3', '64x48+pd3', '64x64', '64x64+pd3', '64x96', '64x96+pd3', '64x128', '64x128+pd3',
                 '128x64+pd3', '32x64', '32x64+pd3'):
        d = _lib.ShapyConv()
        d.in_, d.wgt, d.res, d.out = x.data_ptr(), w.data_ptr(), res.data_ptr(), out.data_ptr()
        d.B, d.Hi, d.Wi, d.Ho, d.Wo, d.Cin, d.in_ld, d.Cout = B, 1, 1, 1, 1, K, K, N
        d.ksize, d.stride, d.pad, d.out_ld, d.res_ld, d.ups = 1, 1, 0, N, N, 1
        d.tile = _lib.TILES[tile]
        d.dtype = _lib.DTYPE_F32
        out.zero_()
        rc = lib.shapy_conv2d(ctypes.byref(d), stream)
        if rc != 0:
            print(f'{tile:12s} rc {rc}')
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        err = (out - ref).abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            lib.shapy_conv2d(ctypes.byref(d), stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        print(f'{tile:12s} {us:7.1f} us  {4.0 * N * K / us / 1e6:6.2f} TB/s of weights  max diff vs first {err:.1e}')


if __name__ == '__main__':
    main()
