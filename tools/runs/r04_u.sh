#!/bin/bash
# GPU run U of round 4: filter refills spread between the MFMAs of a pair (per-layer F(4x4) kernel), A/B
set -u
mkdir -p gpurun_out/r04u
O=gpurun_out/r04u
for v in "" variants/libspread.so variants/libspreadpin.so; do
  echo "=== lib ${v:-product}"
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v; else unset SHAPY_HIP_LIB; fi
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_kernel or winograd4_concat" 2>&1 | tail -1
  timeout 300 python tools/conv_bench.py --tiles wino4 --iters 20 2>&1 | grep -E "wino4" | grep "r1\|256->" | cut -c1-100
  for rep in 1 2; do
    echo "bench: $(timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone")')"
  done
done 2>&1 | tee $O/spread_ab.txt
