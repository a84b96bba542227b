#!/bin/bash
# round 3 run X: event-driven plan (default) -- A/B of the no-barrier variant, the full GPU suite, timeline,
# PMC traffic of the shipped default, final bench line with cpu baseline
set -u
mkdir -p gpurun_out
for v in 0 1 0 1; do echo "no_barriers=$v: $(SHAPY_DAG_NO_BARRIERS=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"; done
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/x_gpu_tests.log 2>&1; tail -4 gpurun_out/x_gpu_tests.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/x_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 200 python tools/timeline.py gpurun_out/x_trace > gpurun_out/x_timeline.txt 2>&1; tail -12 gpurun_out/x_timeline.txt
timeout 200 python tools/timeline.py gpurun_out/x_trace --verbose > gpurun_out/x_timeline_verbose.txt 2>&1
f=$(find gpurun_out/x_trace -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/x_kernel_stats_multistream.csv
rm -rf gpurun_out/x_trace
timeout 900 bash tools/pmc_hbm_traffic.sh gpurun_out/x_pmc_hbm_traffic_winograd4 f32 winograd4 | grep -A8 hbm_bytes
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/x_bench.json 2>gpurun_out/x_bench.err; cat gpurun_out/x_bench.json | cut -c1-2500; tail -2 gpurun_out/x_bench.err
