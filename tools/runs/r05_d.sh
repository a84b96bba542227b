#!/bin/bash
# Prepared at the end of round 4: the 64-channel N tile of the F(4x4) kernel (four multiplying waves, all four SIMDs of a
# CU) on the 192- and 384-channel layers (192 = 3 x 64, 384 = 6 x 64; tile flag 0x400000, SHAPY_WINO4_N64_COUT).
# In the stage-4 branch phases the chip runs at 82-85 % of what THREE multiplying SIMDs per CU can do.
set -u
mkdir -p gpurun_out/r05d
O=gpurun_out/r05d
timeout 300 python -m pytest tests/test_gpu_parity.py -q -rA -k "forced_64_channel" 2>&1 | tail -8 | tee $O/n64_tests.txt
bench1() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", "betas", (d.get("parity") or {}).get("betas_l2"))'; }
for rep in 1 2 3; do
  echo "rep $rep N = 48 everywhere: $(bench1)"
  echo "rep $rep N = 64 for 384: $(SHAPY_WINO4_N64_COUT=384 bench1)"
  echo "rep $rep N = 64 for 192, 384: $(SHAPY_WINO4_N64_COUT=192,384 bench1)"
done 2>&1 | tee $O/n64_bench.txt
