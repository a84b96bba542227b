#!/bin/bash
# GPU run T of round 4: layer1's 64 -> 64 convs on the 64-channel F(4x4) tile vs F(2x2), interleaved A/B
set -u
mkdir -p gpurun_out/r04t
for rep in 1 2 3; do for v in 0 1; do
  echo "rep $rep SHAPY_WINO4_N64=$v: $(SHAPY_WINO4_N64=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone, frac", round(d["roofline"]["frac"],4))')"
done; done | tee gpurun_out/r04t/n64_layer1_ab.txt
