#!/bin/bash
set -u
mkdir -p gpurun_out
for B in 1 8 32; do
  for G in 1 0; do
    echo "B=$B group=$G: $(SHAPY_GROUP_BRANCHES=$G timeout 200 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grouped or conv2d_group" 2>&1 | tail -3
echo "== LBVH workload"
timeout 300 python bench.py --workload bvh --meshes 1000 --steps 10 --warmup 2 > gpurun_out/n_bench_bvh.json 2>gpurun_out/n_bench_bvh.err; cat gpurun_out/n_bench_bvh.json | cut -c1-1500; tail -3 gpurun_out/n_bench_bvh.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/n_bvh_prof -- python $GRAFT_REPO_ROOT/bench.py --workload bvh --meshes 1000 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/n_bvh_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/n_kernel_stats_bvh.csv; head -12 gpurun_out/n_kernel_stats_bvh.csv | cut -c1-200
rm -rf gpurun_out/n_bvh_prof
