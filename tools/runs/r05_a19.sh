#!/bin/bash
# round 5, GPU run 19: the head's 1x1 GEMMs on the bf16x6 kernel (float32 tensors, exact 3-way bf16 split) vs the f32 kernel
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a19 && export TMPDIR=/tmp
for f in 7,2048,2048,1 7,2048,512,1 7,512,2048,1 7,1536,2048,1; do
  timeout 120 python tools/conv_bench.py --dtype f32x6 --filter $f --tiles 64x64,128x64,64x96,128x96,64x128,128x128 --iters 20 2>&1 | grep -v amdgpu.ids | tail -n 1
  timeout 120 python tools/conv_bench.py --dtype f32 --filter $f --tiles auto,32x64 --iters 20 2>&1 | grep -v amdgpu.ids | tail -n 1
done > gpurun_out/a19/head_gemm_x6_vs_f32.txt 2>&1
cat gpurun_out/a19/head_gemm_x6_vs_f32.txt
