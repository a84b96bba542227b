#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 300 python tools/wino4g_check.py --canary --bench 2>&1 | grep -v amdgpu.ids | grep -v "^ok"
