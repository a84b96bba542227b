#!/bin/bash
# GPU run G of round 4: phase stamps of one workgroup of the persistent F(4x4) kernel with the 16-byte epilogue
set -u
mkdir -p gpurun_out/r04g
timeout 600 python tools/wino4g_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04g/w4g_phase_stamps.txt
