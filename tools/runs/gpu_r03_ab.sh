#!/bin/bash
set -u
for B in 1 4 8; do for g in off on; do echo "f32 b$B graph=$g: $(timeout 200 python bench.py --batch $B --graph $g --steps 40 --warmup 8 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c90-200)"; done; done
