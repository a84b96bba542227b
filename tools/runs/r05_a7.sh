#!/bin/bash
# GPU run 7 of round 5: the predicates templated on the scalar type (float32 results must stay bit-exact: every
# operator / measurement / LBVH test) + the float64 operator's first run; skin kernel with the 2 x 2 block.
set -u
mkdir -p gpurun_out/r05a7
O=$GRAFT_REPO_ROOT/gpurun_out/r05a7
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_metrics.py -q -k "mesh or measure or bvh or hull or shipped or smplx or full_forward" 2>&1 | tail -6 | tee $O/tests.txt
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
timeout 300 python bench.py --workload measurements --steps 10 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('measurements', round(d['ms_per_step'],4), 'ms per 1000', d.get('parity'))"
