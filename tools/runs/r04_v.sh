#!/bin/bash
# GPU run V of round 4: the spread-refill variant after its first-chunk fix: tests, classes, interleaved end to end
set -u
mkdir -p gpurun_out/r04v
O=gpurun_out/r04v
export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/variants/libspread.so
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_kernel or winograd4_concat or winograd4_features or features_256" 2>&1 | tail -2
timeout 300 python tools/conv_bench.py --tiles wino4 --iters 20 2>&1 | grep -E "wino4" | grep "r1\|256->" | cut -c1-100
unset SHAPY_HIP_LIB
for rep in 1 2 3; do for v in "" variants/libspread.so; do
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v; else unset SHAPY_HIP_LIB; fi
  echo "rep $rep ${v:-product}: $(timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone")')"
done; done 2>&1 | tee $O/spread_ab.txt
