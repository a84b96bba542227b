#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grouped or conv2d_group or bvh or mesh_to_mesh" 2>&1 | tail -3
echo "== LBVH workload"
timeout 300 python bench.py --workload bvh --meshes 1000 --steps 10 --warmup 2 > gpurun_out/o_bench_bvh.json 2>gpurun_out/o_bench_bvh.err; cat gpurun_out/o_bench_bvh.json | cut -c1-600; tail -3 gpurun_out/o_bench_bvh.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/o_bvh_prof -- python $GRAFT_REPO_ROOT/bench.py --workload bvh --meshes 1000 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/o_bvh_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/o_kernel_stats_bvh.csv; head -5 gpurun_out/o_kernel_stats_bvh.csv | cut -c1-200
rm -rf gpurun_out/o_bvh_prof
