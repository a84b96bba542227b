#!/bin/bash
set -u
for B in 1 8 16; do
echo "f32 b$B multi-stream dag: $(timeout 200 python bench.py --batch $B --steps 40 --warmup 8 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-200)"
echo "f32 b$B single-stream grouped: $(timeout 200 python bench.py --batch $B --single-stream --steps 40 --warmup 8 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-200)"
echo "f32 b$B multi-stream grouped: $(SHAPY_GROUP_BRANCHES=1 timeout 200 python bench.py --batch $B --steps 40 --warmup 8 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-200)"
done
