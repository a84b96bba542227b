"""Phase timing of one conv_wino_kernel workgroup (tuning only): builds a variant library with
-DSHAPY_WINO_TIMING -DSHAPY_MEASURE_TIMING and prints, for a few HRNet layer classes at B = 64,
the wall_clock64 stamps of a mid-grid workgroup: start, first chunk staged, after chunk 0,
K loop done, accumulators parked, stores issued.

    python tools/wino_timing.py            # on a GPU box
"""
import ctypes
import os
import os.path as osp
import subprocess
import sys

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = '/tmp/libshapy_timing.so'


def main():
    env = dict(os.environ, SHAPY_HIPCC_FLAGS='-DSHAPY_WINO_TIMING -DSHAPY_MEASURE_TIMING',
               SHAPY_HIP_LIB=VARIANT)
    subprocess.check_call([sys.executable, '-m', 'shapy_amd.build'], cwd=ROOT, env=env,
                          stdout=subprocess.DEVNULL)
    os.environ['SHAPY_HIP_LIB'] = VARIANT
    import torch
    from shapy_amd import _lib
    from shapy_amd.utils import winograd
    lib = _lib.load()
    lib.shapy_debug_wino_times.restype = ctypes.c_int
    lib.shapy_debug_wino_times.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    B = 64
    cases = [(56, 48, 48, True, 1, 0), (56, 48, 48, True, 1, 0x20000), (28, 96, 96, True, 1, 0),
             (28, 96, 96, True, 2, 0), (14, 192, 192, True, 2, 0), (14, 192, 192, True, 1, 0),
             (7, 384, 384, True, 1, 0)]
    # ablations (timing build only, results are wrong): 1 = filters from one hot 3 KB, 2 = every
    # workgroup loads tile group 0, 4 = no output stores, 8 = a quarter of the MFMAs, 16 = no filter refills (chunk 0's fragments reused)
    dbgs = [int(v) for v in os.environ.get('WINO_DBGS', '0').split(',')]
    for (H, Cin, Cout, res, tm, extra), dbg in [(c, g) for c in cases for g in dbgs]:
        os.environ['SHAPY_WINO_DBG'] = str(dbg)
        x = torch.randn(B, H, H, Cin, device='cuda')
        w = torch.randn(Cout, 3, 3, Cin, device='cuda') * 0.05
        b = torch.randn(Cout, device='cuda')
        out = torch.empty(B, H, H, Cout, device='cuda')
        r = torch.randn_like(out) if res else None
        wu = torch.from_numpy(winograd.transform_filters(w.cpu().numpy())).cuda()
        d = _lib.ShapyConv()
        d.in_ = x.data_ptr(); d.wgt = w.data_ptr(); d.bias = b.data_ptr(); d.out = out.data_ptr()
        d.res = r.data_ptr() if res else None
        d.B, d.Hi, d.Wi, d.Cin, d.in_ld = B, H, H, Cin, Cin
        d.Ho, d.Wo, d.Cout = H, H, Cout
        d.ksize, d.stride, d.pad = 3, 1, 1
        d.out_ld = Cout; d.res_ld = Cout if res else 0; d.relu = 1; d.ups = 1
        d.tile = 0x4000 * tm | extra; d.dtype = _lib.DTYPE_F32; d.wgt_wino = wu.data_ptr()
        for _ in range(3):
            assert lib.shapy_conv2d(ctypes.byref(d), stream) == 0
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.shapy_conv2d(ctypes.byref(d), stream)
        e1.record(); torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * 16)()
        assert lib.shapy_debug_wino_times(buf) == 0
        t = [(buf[i] - buf[0]) / 100.0 for i in range(6)]
        print(f'{H:3d}x{H:<3d} {Cin:3d}->{Cout:<3d} tm={tm} res={int(res)} {hex(extra)} dbg={dbg:2d}  launch {e0.elapsed_time(e1) * 100:.1f} us | '
              f'staged {t[1]:.2f}  chunk0 done {t[2]:.2f}  K loop done {t[3]:.2f}  parked {t[4]:.2f}  '
              f'end {t[5]:.2f} us   (chunks: {Cin // 16})')


if __name__ == '__main__':
    main()
