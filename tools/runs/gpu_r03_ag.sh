#!/bin/bash
# round 3: per-class conv table of the shipped kernels + the configs[3] / SMPL-X lines on the final build
set -u
mkdir -p gpurun_out
timeout 300 python tools/conv_bench.py --tiles auto,wino,wino4 --wino4-min-hw 7 --iters 10 > gpurun_out/ag_conv_bench_default.txt 2>&1; grep -v amdgpu.ids gpurun_out/ag_conv_bench_default.txt | tail -50 | cut -c1-110
timeout 200 python bench.py --workload measurements --meshes 1000 > gpurun_out/ag_bench_measurements_1000.json 2>/dev/null; cut -c1-300 gpurun_out/ag_bench_measurements_1000.json
timeout 200 python bench.py --workload smplx --batch 64 > gpurun_out/ag_bench_smplx_b64.json 2>/dev/null; cut -c1-260 gpurun_out/ag_bench_smplx_b64.json
