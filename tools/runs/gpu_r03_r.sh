#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== benign synthetic weights"; timeout 300 python tools/wino_guard_report.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== wild weights"; timeout 300 python tools/wino_guard_report.py --wild 2>&1 | grep -v amdgpu.ids | tail -30
