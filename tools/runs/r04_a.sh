#!/bin/bash
# GPU run A of round 4: the transposed-accumulator F(4x4) epilogue (16-byte stores) -- canary, class
# timings, grouped launches, end-to-end A/B, --force-gather rehearsal, full suite, full bench line.
set -u
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
export PYTHONUNBUFFERED=1
echo "== canary (kernel-level F(4x4) tests)"
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_kernel or conv2d_group or winograd4_concat or grouped" 2>&1 | tail -3
echo "== conv_bench wino4 classes"
timeout 300 python tools/conv_bench.py --tiles wino,wino4 --iters 20 2>&1 | grep -E "wino4|^#" | tee $O/conv_bench_wino4.txt
echo "== bench A/B (no cpu baseline, no also)"
for v in "" "--group-branches on" "--single-stream" "--single-stream --group-branches on"; do
  echo "bench $v: $(timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also $v 2>&1 | grep -v amdgpu.ids | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone, frac", round(d["roofline"]["frac"],4))')"
done
echo "== grouped launches"
timeout 300 python tools/wino4g_check.py --bench 2>&1 | grep -v amdgpu.ids | tee $O/w4g_bench.txt
echo "== force-gather rehearsal"
for m in work side; do
  echo "force-gather $m: $(timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also --force-gather --gather-mode $m 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bench_force_gather_$m.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", d.get("rccl_ranks"), d.get("force_gather",{}).get("mode"))')"
done
echo "== full GPU suite"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/gpu_tests_tail.txt
echo "== full bench line"
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_default.json
python -c 'import json; d=json.load(open("gpurun_out/r04a/bench_default.json")); print(d["value"], d["roofline"]["frac"], d["parity"]["betas_l2"], d["cpu_baseline"]["value"]); print(json.dumps({k:(v.get("value"), v.get("roofline",{}).get("frac"), v.get("error")) for k,v in d["also"].items()}, indent=1))'
