#!/bin/bash
# GPU run 4 of round 5: the whole GPU suite on the cleaned-up tree (ABI 8), the SMPL-X layer after its round-5
# changes (skin kernel: 4 bodies per thread + W prefetch; pose kernel: batched joint-regression loads; one shape
# GEMM with M = 2B; strided pose parts, leaner host path), bench lines of configs[0] at B = 4 / 64 with a kernel trace.
set -u
mkdir -p gpurun_out/r05a4
O=$GRAFT_REPO_ROOT/gpurun_out/r05a4
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee $O/gpu_tests_tail.txt
for b in 4 64; do
  timeout 300 python bench.py --workload smplx --batch $b --steps 200 --warmup 20 2>/dev/null | grep '^{' | tail -1 > $O/bench_smplx_b$b.json
  python -c "
import json; d=json.load(open('$O/bench_smplx_b$b.json')); r=d['roofline']; print('smplx B=$b', round(d['value']), 'bodies/s', round(r['ms_per_launch_group']*1e3,1), 'us per call', 'frac', round(r['frac'],3))"
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --workload smplx --batch 64 --steps 50 --warmup 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_smplx_b64.csv 2>/dev/null; head -12 $O/kernel_stats_smplx_b64.csv | cut -c1-160
rm -rf $O/prof
