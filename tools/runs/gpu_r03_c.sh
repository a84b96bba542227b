#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/b_trace_ms -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/b_trace_ms.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/b_trace_ss -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --single-stream > $R/gpurun_out/b_trace_ss.log 2>&1
cd $R
echo "== timeline multi-stream"
timeout 200 python tools/timeline.py gpurun_out/b_trace_ms > gpurun_out/b_timeline_ms.txt 2>&1; tail -14 gpurun_out/b_timeline_ms.txt
echo "== timeline single-stream"
timeout 200 python tools/timeline.py gpurun_out/b_trace_ss > gpurun_out/b_timeline_ss.txt 2>&1; tail -14 gpurun_out/b_timeline_ss.txt
timeout 200 python tools/timeline.py gpurun_out/b_trace_ms --verbose > gpurun_out/b_timeline_ms_verbose.txt 2>&1
rm -rf gpurun_out/b_trace_ms gpurun_out/b_trace_ss
