#!/bin/bash
set -u
mkdir -p gpurun_out
SHAPY_W4G_STATIC=1 timeout 400 python tools/wino4g_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/k_stamps_static.txt; head -42 gpurun_out/k_stamps_static.txt
