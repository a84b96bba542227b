#!/bin/bash
# round 3 closing run: smoke, full GPU suite, headline bench (+ rocprofv3 kernel stats of the same command), LBVH bench
set -u
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1; tail -4 gpurun_out/final_gpu_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench.json 2>gpurun_out/final_bench.err; cut -c1-700 gpurun_out/final_bench.json; tail -2 gpurun_out/final_bench.err
timeout 300 python bench.py --workload bvh --meshes 1000 --steps 10 --warmup 2 > gpurun_out/final_bench_bvh.json 2>gpurun_out/final_bench_bvh.err; cut -c1-330 gpurun_out/final_bench_bvh.json; python -c "
import json; r=json.loads(open('gpurun_out/final_bench_bvh.json').read().strip().splitlines()[-1]); print(r['parity'], r['overflowed_hits'], r['hits_per_pair'])"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/final_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/final_kernel_stats_multistream.csv; head -8 gpurun_out/final_kernel_stats_multistream.csv | cut -c1-160
rm -rf gpurun_out/final_prof
