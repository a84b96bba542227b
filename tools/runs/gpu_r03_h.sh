#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== per-layer kernel with the 3-deep residual epilogue (was 51-52 / 40 / 41 / 179 us)"
timeout 300 python tools/conv_bench.py --tiles wino4 --wino4-min-hw 7 --iters 10 2>&1 | grep wino4
echo "== grouped, static"; SHAPY_W4G_STATIC=1 timeout 300 python tools/wino4g_check.py --canary --bench 2>&1 | grep -v amdgpu.ids | grep -v "^ok"
echo "== stamps, static"; SHAPY_W4G_STATIC=1 timeout 400 python tools/wino4g_timing.py 2>&1 | grep -v amdgpu.ids | head -45
