#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== dynamic (atomic tickets)"; timeout 300 python tools/wino4g_check.py --bench 2>&1 | grep -v amdgpu.ids
echo "== static round-robin"; SHAPY_W4G_STATIC=1 timeout 300 python tools/wino4g_check.py --canary --bench 2>&1 | grep -v amdgpu.ids | grep -v "^ok"
echo "== B=32 dynamic"; timeout 300 python tools/wino4g_check.py --bench --batch 32 2>&1 | grep -v amdgpu.ids
echo "== B=32 static"; SHAPY_W4G_STATIC=1 timeout 300 python tools/wino4g_check.py --bench --batch 32 2>&1 | grep -v amdgpu.ids
