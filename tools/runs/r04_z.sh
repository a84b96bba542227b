#!/bin/bash
# GPU run Z of round 4: the default bench line twice more on another box (run Y's 4,982 against 5,110-5,129 earlier)
set -u
mkdir -p gpurun_out/r04z
for i in 1 2; do
  timeout 200 python bench.py --no-also 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04z/bench_default_$i.json
  python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("ms_per_launch_group"))' gpurun_out/r04z/bench_default_$i.json
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
