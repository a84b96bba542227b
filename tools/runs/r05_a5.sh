#!/bin/bash
# GPU run 5 of round 5: the GPU suite to its end (run 4 stopped at the grouped-vs-split comparison, fixed), SMPL-X
# layer with the 4 x 2 register-blocked skin kernel, tile sweep of the skinny pose-blend GEMM, kernel stats.
set -u
mkdir -p gpurun_out/r05a5
O=$GRAFT_REPO_ROOT/gpurun_out/r05a5
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/gpu_tests_tail.txt
for b in 4 64; do
  timeout 300 python bench.py --workload smplx --batch $b --steps 200 --warmup 20 2>/dev/null | grep '^{' | tail -1 > $O/bench_smplx_b$b.json
  python -c "
import json; d=json.load(open('$O/bench_smplx_b$b.json')); r=d['roofline']; print('smplx B=$b', round(d['value']), 'bodies/s', round(r['ms_per_launch_group']*1e3,1), 'us per call', 'frac', round(r['frac'],3))"
done | tee $O/smplx.txt
timeout 200 python tools/skinny_gemm_bench.py --batch 64 2>&1 | tee $O/skinny_gemm_b64.txt
timeout 200 python tools/skinny_gemm_bench.py --batch 4 2>&1 | tee $O/skinny_gemm_b4.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --workload smplx --batch 64 --steps 50 --warmup 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_smplx_b64.csv 2>/dev/null; grep "shapy" $O/kernel_stats_smplx_b64.csv | cut -c1-150
rm -rf $O/prof
