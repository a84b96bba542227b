#!/bin/bash
set -u
for g in off on; do echo "bf16 b32 graph=$g: $(timeout 200 python bench.py --dtype bf16 --batch 32 --graph $g --steps 30 --warmup 6 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-190)"; done
for g in off on; do echo "bf16 b64 graph=$g: $(timeout 200 python bench.py --dtype bf16 --batch 64 --graph $g --steps 30 --warmup 6 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-190)"; done
echo "bf16 b32 dag off: $(SHAPY_DAG=0 timeout 200 python bench.py --dtype bf16 --batch 32 --steps 30 --warmup 6 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-190)"
echo "f32 b64 graph on: $(timeout 200 python bench.py --graph on --steps 20 --warmup 6 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-190)"
