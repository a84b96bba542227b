#!/bin/bash
# GPU run C of round 4: which change removes the error of the transposed-accumulator F(4x4) kernel?
set -u
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
for v in variants/libpin.so variants/libdrain.so variants/libboth.so variants/libaf3keep.so variants/libloopnop.so; do
  echo "=== lib $v"
  export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v
  timeout 200 python tools/w4_debug.py 2>&1 | grep -v amdgpu.ids | grep -v "bad by channel\|best matching"
done | tee $O/w4_debug.txt
