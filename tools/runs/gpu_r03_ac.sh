#!/bin/bash
set -u
for B in 1 8; do for g in off on; do echo "f32 b$B graph=$g: $(timeout 200 python bench.py --batch $B --graph $g --steps 40 --warmup 8 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-200)"; done; done
echo "f32 b1 graph=on no prio: $(SHAPY_LANE_PRIO=0 timeout 200 python bench.py --batch 1 --graph on --steps 40 --warmup 8 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-200)"
echo "f32 b64: $(timeout 200 python bench.py --steps 20 --warmup 6 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-200)"
echo "f32 b64 graph on: $(timeout 200 python bench.py --graph on --steps 20 --warmup 6 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-200)"
