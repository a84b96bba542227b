#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "event_driven or grouped" 2>&1 | tail -4
for d in 1 0 1 0; do echo "dag=$d: $(SHAPY_DAG=$d timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"; done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/u_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 200 python tools/timeline.py gpurun_out/u_trace > gpurun_out/u_timeline_dag.txt 2>&1; tail -14 gpurun_out/u_timeline_dag.txt
timeout 200 python tools/timeline.py gpurun_out/u_trace --verbose > gpurun_out/u_timeline_dag_verbose.txt 2>&1
rm -rf gpurun_out/u_trace
