#!/bin/bash
# round 2, GPU run Y (the last ~60 GPU-seconds): the one GPU test added after run X, and the
# unmeasured 12-chunk-unrolled F(4x4) variant (parity against the generic loop + per-class time)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 40 python -m pytest -q -x --tb=short tests/test_gpu_parity.py \
    -k "falls_back_to_direct or (winograd4_kernel and 192)" > gpurun_out/y_tests.log 2>&1
tail -n 2 gpurun_out/y_tests.log
timeout 40 python - > gpurun_out/y_unroll12.txt 2>&1 <<'PY'
import sys
import numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_parity as t
from shapy_amd import _lib
lib = _lib.load()
g = torch.Generator().manual_seed(1)
for B in (1, 64):
    x = torch.randn(B, 14, 14, 192, generator=g).cuda()
    w = (torch.randn(192, 3, 3, 192, generator=g) / np.sqrt(9 * 192)).cuda()
    b = torch.randn(192, generator=g).cuda()
    r = torch.randn(B, 14, 14, 192, generator=g).cuda()
    a = t._conv_call(lib, x, w, b, r, True, 1, 1, wino=4)
    u = t._conv_call(lib, x, w, b, r, True, 1, 1, wino=4, tile=_lib.TILE_WINO4_UNROLL12)
    print('B', B, 'unrolled == generic loop:', torch.equal(a, u))
    if B == 1:
        print('err vs f64', (u.cpu().double() - t._conv_ref(x, w, b, r, True, 1, 1)).abs().max().item())
def timeit(flag, n=30):
    for _ in range(3):
        t._conv_call(lib, x, w, b, r, True, 1, 1, wino=4, tile=flag)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    import ctypes
    from shapy_amd.utils import winograd
    d = _lib.ShapyConv()
    wu = torch.from_numpy(winograd.transform_filters4(w.cpu().numpy())).cuda()
    out = torch.empty_like(r)
    d.dtype = _lib.DTYPE_F32; d.in_ = x.data_ptr(); d.wgt = w.data_ptr(); d.bias = b.data_ptr()
    d.res = r.data_ptr(); d.out = out.data_ptr(); d.wgt_wino = wu.data_ptr()
    d.B, d.Hi, d.Wi, d.Cin, d.in_ld = 64, 14, 14, 192, 192
    d.Ho, d.Wo, d.Cout = 14, 14, 192
    d.ksize, d.stride, d.pad = 3, 1, 1
    d.out_ld = 192; d.res_ld = 192; d.relu = 1; d.ups = 1; d.tile = _lib.TILE_WINO4 | flag
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(5):
        lib.shapy_conv2d(ctypes.byref(d), s)
    e0.record()
    for _ in range(n):
        lib.shapy_conv2d(ctypes.byref(d), s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for rep in range(2):
    print('192->192 @14x14 B=64: generic loop %.1f us, unrolled %.1f us' % (timeit(0), timeit(_lib.TILE_WINO4_UNROLL12)))
PY
cat gpurun_out/y_unroll12.txt | tail -n 6
