#!/bin/bash
# round 3, first GPU run (prepared at the end of round 2, which ran out of GPU minutes right after the
# F(4x4) Winograd kernel became the default): the PMC passes that build still owes, then the A/B of
# the knob that was added unmeasured.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== PMC: HBM bytes + MFMA busy cycles of the F(4x4) default (roofline.traffic)"
timeout 900 bash tools/pmc_hbm_traffic.sh gpurun_out/a_pmc_hbm_traffic_winograd4 f32 winograd4 | grep -A8 hbm_bytes
echo "== F(4x4) kernel: 12-chunk loop unrolled for Cin = 192 (tile flag 0x200000; round 2 run Y: 40.9 vs 41.3 us, no gain)"
timeout 200 python tools/conv_bench.py --tiles wino4,wino4u12 --filter 14,192,192,3 --iters 20 | grep wino4
timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline | cut -c90-200
timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --tile-flags 0x200000 | cut -c90-200
echo "== parity of the unrolled variant (Cin = 192 case of the kernel test runs the default loop: check by hand)"
timeout 300 python - <<'PY'
import ctypes, numpy as np, torch, sys
sys.path.insert(0, '.')
from tests.test_gpu_parity import _conv_call, _conv_ref
from shapy_amd import _lib
lib = _lib.load()
g = torch.Generator().manual_seed(1)
x = torch.randn(2, 14, 14, 192, generator=g).cuda()
w = (torch.randn(192, 3, 3, 192, generator=g) / np.sqrt(9 * 192)).cuda()
b = torch.randn(192, generator=g).cuda()
a = _conv_call(lib, x, w, b, None, True, 1, 1, wino=4)
u = _conv_call(lib, x, w, b, None, True, 1, 1, wino=4, tile=_lib.TILE_WINO4_UNROLL12)
print('unrolled == generic loop:', torch.equal(a, u), 'err vs f64', (u.cpu().double() - _conv_ref(x, w, b, None, True, 1, 1)).abs().max().item())
PY
echo "== EXPERIMENTAL half-position F(4x4) kernel (csrc/conv_wino4h.hip, tile flag 0x400000): canary under a timeout"
timeout 120 python - <<'PY'
import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_parity as t
from shapy_amd import _lib
lib = _lib.load()
g = torch.Generator().manual_seed(2)
for (B, H, W, C, O, use_res) in ((1, 8, 8, 16, 48, False), (2, 12, 20, 48, 48, True), (1, 7, 9, 32, 96, True),
                                 (2, 28, 28, 96, 96, True), (2, 56, 56, 48, 48, True), (1, 14, 14, 192, 192, True)):
    x = torch.randn(B, H, W, C, generator=g).cuda()
    w = (torch.randn(O, 3, 3, C, generator=g) / np.sqrt(9 * C)).cuda()
    b = torch.randn(O, generator=g).cuda()
    r = torch.randn(B, H, W, O, generator=g).cuda() if use_res else None
    a = t._conv_call(lib, x, w, b, r, True, 1, 1, wino=4)
    hh = t._conv_call(lib, x, w, b, r, True, 1, 1, wino=4, tile=_lib.TILE_WINO4_HALF)
    ref = t._conv_ref(x, w, b, r, True, 1, 1)
    print((B, H, W, C, O), 'half vs full kernel', (a - hh).abs().max().item(),
          'half vs f64', (hh.cpu().double() - ref).abs().max().item())
PY
echo "== half-position kernel: per-class timings next to the default kernel"
timeout 200 python tools/conv_bench.py --tiles wino,wino4,wino4h --wino4-min-hw 7 --iters 10 | grep wino4
timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --tile-flags 0x400000 | cut -c90-200
