#!/bin/bash
# GPU run 21 of round 5 (closing build): the whole GPU suite, smoke, the default bench with its wall time, rocprofv3
# kernel stats of the default (pipelined) command.
set -u
mkdir -p gpurun_out/r05a21
O=$GRAFT_REPO_ROOT/gpurun_out/r05a21
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/gpu_tests_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -4 | tee $O/smoke.txt
t0=$(date +%s.%N)
timeout 600 python bench.py > $O/bench_default.out 2> $O/bench_default.err
t1=$(date +%s.%N)
grep '^{' $O/bench_default.out | tail -1 > $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default bench:', round(d['value'],1), 'img/s', 'frac', round(d['roofline']['frac'],4), 'wall', round($t1-$t0,1), 's'); print({k: round(v,1) for k, v in d.items() if k.startswith('also_') and k.endswith('_value')}); print(d.get('parity')); print(d.get('cpu_baseline'))" | tee $O/bench_default_summary.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-also > $O/bench_under_rocprof.out 2>/dev/null
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_regressor_pipelined.csv
head -8 $O/kernel_stats_regressor_pipelined.csv | cut -c1-150
rm -rf $O/prof
