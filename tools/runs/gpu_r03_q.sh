#!/bin/bash
set -u
mkdir -p gpurun_out
export SHAPY_HIPCC_FLAGS='-DSHAPY_W4G_TIMING' SHAPY_HIP_LIB=/tmp/libshapy_w4g_abl.so
python -m shapy_amd.build > /dev/null 2>&1
for dbg in 0 1 2 3; do
  echo "== grouped kernel ablation dbg=$dbg (1 = no stores, 2 = no residual loads)"
  SHAPY_W4G_DBG=$dbg timeout 200 python tools/wino4g_check.py --bench 2>&1 | grep -v amdgpu.ids | cut -c1-90
done
unset SHAPY_HIPCC_FLAGS SHAPY_HIP_LIB
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 200 python tools/timeline.py gpurun_out/q_trace > gpurun_out/q_timeline_grouped.txt 2>&1; tail -45 gpurun_out/q_timeline_grouped.txt
timeout 200 python tools/timeline.py gpurun_out/q_trace --verbose > gpurun_out/q_timeline_grouped_verbose.txt 2>&1
rm -rf gpurun_out/q_trace
