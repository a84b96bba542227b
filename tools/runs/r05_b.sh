#!/bin/bash
# Prepared at the end of round 4: the per-CU matrix-pipe token of the F(4x4) kernel (-DSHAPY_W4_TOKEN=<max chunks>,
# csrc/conv_wino4.hip).  Build the variants first (CPU, ~1 min each):
#   for n in 3 6; do SHAPY_HIPCC_FLAGS="-DSHAPY_W4_TOKEN=$n" SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/variants/libtok$n.so python -m shapy_amd.build; done
# Model (profiles/r04w_*, r04ab_*): two resident workgroups multiply in phase and finish together; with the
# multiply phases serialised a CU with three 48-channel tasks takes ~38 instead of ~53 us.  Isolated targets: 48@56^2
# 50 -> <= 42 us; 96@28^2 (0.77 rounds) unchanged.
set -u
mkdir -p gpurun_out/r05b
O=gpurun_out/r05b
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
