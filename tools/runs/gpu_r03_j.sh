#!/bin/bash
# round 3 run J: F(4x4) kernels without scratch (every kernel that touches scratch pays ~17 us per launch)
set -u
mkdir -p gpurun_out
echo "== per-layer kernel, no scratch (was 51-52 / 40 / 41 / 69 / 179 us)"
timeout 300 python tools/conv_bench.py --tiles wino,wino4 --wino4-min-hw 7 --iters 10 2>&1 | grep wino4
echo "== grouped static"; SHAPY_W4G_STATIC=1 timeout 300 python tools/wino4g_check.py --canary --bench 2>&1 | grep -v amdgpu.ids | grep -v "^ok"
echo "== grouped dynamic"; timeout 300 python tools/wino4g_check.py --bench 2>&1 | grep -v amdgpu.ids
echo "== bench"
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/j_bench.json 2>/dev/null; cut -c1-330 gpurun_out/j_bench.json
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --wino4-min-hw 7 | cut -c90-330
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --single-stream | cut -c90-330
echo "== wino4 GPU tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wino or hrnet or full_forward" 2>&1 | tail -4
