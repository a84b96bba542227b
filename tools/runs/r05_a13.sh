#!/bin/bash
# GPU run 13 of round 5: the whole GPU suite on the closing tree (split-K on both kernel families, batch buckets), smoke,
# step latency by batch, the default bench.
set -u
mkdir -p gpurun_out/r05a13
O=$GRAFT_REPO_ROOT/gpurun_out/r05a13
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/gpu_tests_tail.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -11 | tee $O/smoke.txt
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", round(d["ms_per_step"],3), "ms per step")'; }
for b in 1 4 8 16 32 64; do echo "f32 B=$b: $(bench --batch $b)"; done | tee $O/bench_by_batch.txt
echo "bf16 B=32: $(bench --dtype bf16 --batch 32)" | tee -a $O/bench_by_batch.txt
t0=$(date +%s.%N)
timeout 600 python bench.py > $O/bench_default.out 2> $O/bench_default.err
t1=$(date +%s.%N)
grep '^{' $O/bench_default.out | tail -1 > $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default bench:', round(d['value'],1), 'img/s', 'frac', round(d['roofline']['frac'],4), 'wall', round($t1-$t0,1), 's'); print({k: round(v,1) for k, v in d.items() if k.startswith('also_') and k.endswith('_value')})"
