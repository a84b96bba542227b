#!/bin/bash
set -u
mkdir -p gpurun_out
for n in 1 2 3 2; do echo "pipeline_streams=$n: $(timeout 200 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --pipeline-streams $n 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-190)"; done
echo "pipeline_streams=2 + GPU_MAX_HW_QUEUES=8: $(GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --pipeline-streams 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-190)"
echo "pipeline_streams=2 + dag off: $(SHAPY_DAG=0 timeout 200 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --pipeline-streams 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-190)"
