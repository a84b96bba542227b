#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/final_gpu_tests.log 2>&1; tail -3 gpurun_out/final_gpu_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench.json 2>gpurun_out/final_bench.err; cut -c1-330 gpurun_out/final_bench.json; tail -2 gpurun_out/final_bench.err
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/final_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/final_kernel_stats_multistream.csv; grep "stem_conv" gpurun_out/final_kernel_stats_multistream.csv | cut -c1-200
rm -rf gpurun_out/final_prof
