#!/bin/bash
# GPU run E of round 4: the store-data hazard fix of the 16-byte F(4x4) epilogue (+ the variant that pins
# the V-fragment reads one pair ahead): correctness, class timings, end to end.
set -u
mkdir -p gpurun_out/r04e
O=gpurun_out/r04e
for v in "" variants/libpin.so; do
  echo "=== lib ${v:-product}"
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v; else unset SHAPY_HIP_LIB; fi
  timeout 200 python tools/w4_debug.py 2>&1 | grep -v amdgpu.ids | grep "max err\|bad elements"
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_kernel or conv2d_group or winograd4_concat or grouped" 2>&1 | tail -2
  timeout 300 python tools/conv_bench.py --tiles wino4 --iters 20 2>&1 | grep -E "wino4" | tee $O/conv_bench_wino4_$(basename ${v:-product} .so).txt
  for f in "" "--group-branches on"; do
    echo "bench $f: $(timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also $f 2>&1 | grep -v amdgpu.ids | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone, frac", round(d["roofline"]["frac"],4))')"
  done
done 2>&1 | tee $O/log.txt
