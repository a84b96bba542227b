#!/bin/bash
# Prepared at the end of round 4 for the FIRST GPU run of round 5: the per-CU matrix-pipe token of the F(4x4) kernel
# (-DSHAPY_W4_TOKEN=<max chunks>: conv_wino4.hip).  Build the variants first (CPU, ~1 min each):
#   for n in 3 6; do SHAPY_HIPCC_FLAGS="-DSHAPY_W4_TOKEN=$n" SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/variants/libtok$n.so python -m shapy_amd.build; done
# Model (profiles/r04w_*, r04ab_*): residents in phase 32 us per pair of 48-channel tasks, with the
# multiply phases serialised 21-27 us.  Isolated targets: 48@56^2 50 -> <= 40 us, 96@28^2 39 -> <= 33 us.
set -u
mkdir -p gpurun_out/r05a
O=gpurun_out/r05a
for v in "" variants/libtok3.so variants/libtok6.so; do
  echo "=== lib ${v:-product}"
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v; else unset SHAPY_HIP_LIB; fi
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_kernel or winograd4_concat" 2>&1 | tail -1
  timeout 200 python tools/conv_bench.py --tiles wino4 --iters 20 2>&1 | grep -E "wino4" | grep "r1\|256->" | cut -c1-100
done 2>&1 | tee $O/token_classes.txt
for rep in 1 2 3; do for v in "" variants/libtok3.so variants/libtok6.so; do
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v; else unset SHAPY_HIP_LIB; fi
  echo "rep $rep ${v:-product}: $(timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone")')"
done; done 2>&1 | tee $O/token_bench.txt
# --- second prepared experiment: the fuse_add plans (HighResolutionNet.fuse_add = 1 | 2) ---
unset SHAPY_HIP_LIB
timeout 900 python -m pytest tests/test_zz_fuse_add_gpu.py -q -rA 2>&1 | tail -30 | tee $O/fuse_add_tests.txt
bench1() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone")'; }
for rep in 1 2 3; do
  echo "rep $rep fuse_add 0: $(bench1 --fuse-add 0)"
  echo "rep $rep fuse_add 1: $(bench1 --fuse-add 1)"
  echo "rep $rep fuse_add 2 dest,dest,mixed: $(bench1 --fuse-add 2 --fuse-chain-lanes dest,dest,mixed)"
  echo "rep $rep fuse_add 2 dest,dest,dest: $(bench1 --fuse-add 2 --fuse-chain-lanes dest,dest,dest)"
  echo "rep $rep fuse_add 2 source,source,source: $(bench1 --fuse-add 2 --fuse-chain-lanes source,source,source)"
  echo "rep $rep fuse_add 1 + launch groups: $(bench1 --fuse-add 1 --group-branches on)"
  echo "rep $rep fuse_add 2 + launch groups: $(bench1 --fuse-add 2 --group-branches on)"
done 2>&1 | tee $O/fuse_add_bench.txt
# the winner's timeline: rocprofv3 --kernel-trace --stats --output-format csv ... then tools/timeline.py --verbose
