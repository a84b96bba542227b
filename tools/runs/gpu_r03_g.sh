#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== static"; SHAPY_W4G_STATIC=1 timeout 400 python tools/wino4g_timing.py 2>&1 | grep -v amdgpu.ids
echo "== dynamic"; timeout 400 python tools/wino4g_timing.py 2>&1 | grep -v amdgpu.ids | head -60
