#!/bin/bash
# GPU run 1 of round 5: (a) the one prepared experiment the driver had not run -- the per-CU matrix-pipe token of the
# F(4x4) kernel (-DSHAPY_W4_TOKEN=3|6, variants built on the CPU beforehand) against the product library, per class and
# end to end; (b) the baseline of the closing round-4 build on THIS box for the split-K work: isolated class times.
set -u
mkdir -p gpurun_out/r05a1
O=gpurun_out/r05a1
for v in "" variants/libtok3.so variants/libtok6.so; do
  echo "=== lib ${v:-product}"
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v; else unset SHAPY_HIP_LIB; fi
  timeout 200 python tools/conv_bench.py --tiles wino4 --iters 20 2>&1 | grep -E "wino4" | grep "r1" | cut -c1-110
done 2>&1 | tee $O/token_classes.txt
for rep in 1 2; do for v in "" variants/libtok3.so variants/libtok6.so; do
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v; else unset SHAPY_HIP_LIB; fi
  echo "rep $rep ${v:-product}: $(timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone")')"
done; done 2>&1 | tee $O/token_bench.txt
