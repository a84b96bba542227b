#!/bin/bash
# GPU run B of round 4: localise the error of the transposed-accumulator F(4x4) epilogue (run A:
# every F(4x4) test but the first kernel case failed; the Winograd guard demoted the layers in the bench)
set -u
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
for v in "" variants/libfmax.so variants/libold.so; do
  echo "=== lib ${v:-product}"
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/$v; else unset SHAPY_HIP_LIB; fi
  timeout 200 python tools/w4_debug.py 2>&1 | grep -v amdgpu.ids
done | tee $O/w4_debug.txt
