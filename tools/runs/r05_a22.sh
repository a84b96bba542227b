#!/bin/bash
# GPU run 22 of round 5: ReLU as compare + select in every epilogue (a NaN activation stays NaN): the whole GPU suite with
# the new NaN test, smoke, the default bench.
set -u
mkdir -p gpurun_out/r05a22
O=$GRAFT_REPO_ROOT/gpurun_out/r05a22
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/gpu_tests_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee $O/smoke.txt
t0=$(date +%s.%N)
timeout 600 python bench.py > $O/bench_default.out 2> $O/bench_default.err
t1=$(date +%s.%N)
grep '^{' $O/bench_default.out | tail -1 > $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default bench:', round(d['value'],1), 'img/s', 'frac', round(d['roofline']['frac'],4), 'wall', round($t1-$t0,1), 's'); print({k: round(v,1) for k, v in d.items() if k.startswith('also_') and k.endswith('_value')}); print(d.get('parity'))" | tee $O/bench_default_summary.txt
