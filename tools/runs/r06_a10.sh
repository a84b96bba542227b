#!/bin/bash
# GPU run 10 of round 6: plain-GEMM form of the implicit-GEMM loop for 1x1 layers: class times with / without,
# kernel tests, end to end.
set -u
O=gpurun_out/r06a10
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_kernel_vs_float64 or igemm_split or concat_offset_and_inplace or head_gemms_on_bf16x6 or grouped_branch" 2>&1 | tail -4 | tee $O/tests.txt
timeout 300 python tools/conv_bench.py --tiles auto,auto+nop11 --iters 20 2>&1 | grep -E " k1 s1 u1 " | cut -c1-160 | tee $O/classes_1x1.txt
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms", d["betas_sha1"])'; }
for rep in 1 2; do
  echo "rep $rep p11: $(bench)   unpipelined: $(bench --pipeline off)"
  echo "rep $rep generic: $(SHAPY_TILE_FLAGS=0x1000000 bench)   unpipelined: $(SHAPY_TILE_FLAGS=0x1000000 bench --pipeline off)"
done 2>&1 | tee $O/ab.txt
