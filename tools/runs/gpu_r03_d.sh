#!/bin/bash
# round 3 run D: fork event recorded at the epoch start (hrnet_ops.hip); full GPU suite on the build
# with the ADVICE fixes; bench + timeline
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/d_bench.json 2>gpurun_out/d_bench.err; cut -c1-400 gpurun_out/d_bench.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/d_gpu_tests.log 2>&1; tail -5 gpurun_out/d_gpu_tests.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/d_trace_ms -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/d_trace_ms.log 2>&1
cd $R
timeout 200 python tools/timeline.py gpurun_out/d_trace_ms > gpurun_out/d_timeline_ms.txt 2>&1; tail -12 gpurun_out/d_timeline_ms.txt
timeout 200 python tools/timeline.py gpurun_out/d_trace_ms --verbose > gpurun_out/d_timeline_ms_verbose.txt 2>&1
rm -rf gpurun_out/d_trace_ms
