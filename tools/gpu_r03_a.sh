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
