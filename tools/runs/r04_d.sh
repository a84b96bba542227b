#!/bin/bash
set -u
mkdir -p gpurun_out/r04d
timeout 200 python tools/w4_debug.py 2>&1 | grep -v amdgpu.ids | sed -n '/value forensics/,$p' | tee gpurun_out/r04d/forensics.txt
