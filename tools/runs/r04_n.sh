#!/bin/bash
# GPU run N of round 4: validation of the tree (smoke, full GPU suite, the default bench line)
set -u
mkdir -p gpurun_out/r04n
O=gpurun_out/r04n
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/smoke.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_tests_tail.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python -c 'import json; d=json.load(open("gpurun_out/r04n/bench_default.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity"]["betas_l2"], d["cpu_baseline"]["value"]); print({k:(round(v.get("value",0),1), round(v.get("roofline",{}).get("frac",0),4), v.get("error")) for k,v in d["also"].items()})'
