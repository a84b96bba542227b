#!/bin/bash
# GPU run 6 of round 5: SMPL-X layer with the skin kernel's five-joint weight prefetch and the 32 x 64 tile for the
# pose-blend GEMM (tests, stand-alone bench at B = 4 / 64, kernel stats); wall time of the default bench.py run.
set -u
mkdir -p gpurun_out/r05a6
O=$GRAFT_REPO_ROOT/gpurun_out/r05a6
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "smplx or full_forward or shipped or lut" 2>&1 | tail -4 | tee $O/tests_smplx.txt
for b in 4 64; do
  timeout 300 python bench.py --workload smplx --batch $b --steps 200 --warmup 20 2>/dev/null | grep '^{' | tail -1 > $O/bench_smplx_b$b.json
  python -c "
import json; d=json.load(open('$O/bench_smplx_b$b.json')); r=d['roofline']; print('smplx B=$b', round(d['value']), 'bodies/s', round(r['ms_per_launch_group']*1e3,1), 'us per call', 'frac', round(r['frac'],3))"
done | tee $O/smplx.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --workload smplx --batch 64 --steps 50 --warmup 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_smplx_b64.csv 2>/dev/null; grep "shapy" $O/kernel_stats_smplx_b64.csv | cut -c1-150
rm -rf $O/prof
t0=$(date +%s.%N)
timeout 600 python bench.py > $O/bench_default.out 2> $O/bench_default.err
t1=$(date +%s.%N)
grep '^{' $O/bench_default.out | tail -1 > $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default bench:', round(d['value'],1), 'img/s', 'frac', round(d['roofline']['frac'],4), 'wall', round($t1-$t0,1), 's'); print({k: v for k, v in d.items() if k.startswith('also_') and k.endswith('_value')}); print(d.get('parity')); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'][-80:])"
