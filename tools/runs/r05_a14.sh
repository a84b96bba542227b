#!/bin/bash
# GPU run 14 of round 5: the flat-K bf16 kernel's tail chunk (broken by the chunk counter of run 13's build: six bf16
# tests failed), fixed: the bf16 tests, the GPU suite again, default bench wall time.
set -u
mkdir -p gpurun_out/r05a14
O=$GRAFT_REPO_ROOT/gpurun_out/r05a14
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/gpu_tests_tail.txt
t0=$(date +%s.%N)
timeout 600 python bench.py > $O/bench_default.out 2> $O/bench_default.err
t1=$(date +%s.%N)
grep '^{' $O/bench_default.out | tail -1 > $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default bench:', round(d['value'],1), 'img/s', 'frac', round(d['roofline']['frac'],4), 'wall', round($t1-$t0,1), 's'); print({k: round(v,1) for k, v in d.items() if k.startswith('also_') and k.endswith('_value')})"
