#!/bin/bash
# GPU run I of round 4: full suite + full bench line on the build with the 16-byte F(4x4) epilogue
set -u
mkdir -p gpurun_out/r04i
O=gpurun_out/r04i
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/gpu_tests_tail.txt
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_default.json
python -c 'import json; d=json.load(open("gpurun_out/r04i/bench_default.json")); print(d["value"], d["roofline"]["frac"], d["parity"]["betas_l2"], d["cpu_baseline"]["value"], d["cpu_baseline"]["batch_4"]["value"]); print(json.dumps({k:(v.get("value"), v.get("roofline",{}).get("frac"), v.get("error")) for k,v in d["also"].items()}, indent=1))'
for m in work side; do
  echo "force-gather $m: $(timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also --force-gather --gather-mode $m 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bench_force_gather_$m.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", d.get("rccl_ranks"), d.get("force_gather",{}).get("mode"))')"
done
echo "plain: $(timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also 2>&1 | grep -v amdgpu.ids | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1))')"
