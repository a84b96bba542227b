#!/bin/bash
# GPU run M of round 4: (1) s_setprio 3 for the multiplying waves, A/B; (2) ablations of the per-layer F(4x4)
# kernel on the round-4 build (timing build, wrong results on purpose: which stage costs what per class)
set -u
mkdir -p gpurun_out/r04m
O=gpurun_out/r04m
for v in "" variants/libprio3.so; do
  echo "=== lib ${v:-product}"
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v; else unset SHAPY_HIP_LIB; fi
  timeout 300 python tools/conv_bench.py --tiles wino4 --iters 20 2>&1 | grep -E "wino4" | grep "r1\|256->" | cut -c1-100
  echo "bench: $(timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone")')"
done 2>&1 | tee $O/prio3_ab.txt
export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/variants/libabl.so
for dbg in 0 3 4 8 12 16 32 63; do
  echo "== dbg=$dbg"
  SHAPY_WINO_DBG=$dbg timeout 200 python tools/conv_bench.py --tiles wino4 --wino4-min-hw 7 --iters 10 2>&1 | grep "wino4" | grep "r1\|256->" | cut -c1-90
done 2>&1 | tee $O/wino4_ablations.txt
