#!/bin/bash
# GPU run 19 of round 6 (closing evidence): full GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the
# same command, PMC passes over the four-lane plan (HBM bytes per forward, MFMA busy cycles).
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r06a19
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_tests_tail.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -11 | tee $O/smoke.txt
bash tools/pmc_hbm_traffic.sh gpurun_out/r06a19/pmc_hbm f32 winograd4 64 multi > $O/pmc_tail.txt 2>&1
cp $O/pmc_hbm.json profiles/r06r_pmc_hbm_traffic_winograd4_multilane.json
cp $O/pmc_hbm_per_launch.csv profiles/r06r_pmc_per_launch_winograd4_multilane.csv
( time timeout 900 python bench.py 2>$O/bench_stderr.txt | grep '^{' | tail -1 > $O/bench_default.json ) 2>&1 | grep real | tee $O/bench_wall.txt
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac']); print('mfma_busy', d['roofline'].get('mfma_busy')); print('parity', {k: d['parity'][k] for k in list(d['parity'])[:8]}); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:80])
print({k: round(v, 1) for k, v in d.items() if k.startswith('also_') and k.endswith('_value')})" | tee $O/bench_summary.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-also > $O/bench_under_rocprof.out 2>/dev/null
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_regressor_pipelined.csv
head -8 $O/kernel_stats_regressor_pipelined.csv | cut -c1-150
rm -rf $O/prof
