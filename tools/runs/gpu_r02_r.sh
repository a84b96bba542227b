#!/bin/bash
# Run R: 16-byte epilogue for bf16 and for the upsample-scatter (fuse) layers.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== conv + hrnet tests"
timeout 900 python -m pytest tests -q -m gpu -x -k "conv or hrnet or bf16" 2>&1 | tail -4
echo "== conv bench f32 (ups classes)"
timeout 400 python tools/conv_bench.py --tiles auto > gpurun_out/conv_bench_r02r_f32.txt 2>&1; grep -E " u[248] " gpurun_out/conv_bench_r02r_f32.txt | cut -c1-120
echo "== conv bench bf16 bs32"
timeout 400 python tools/conv_bench.py --dtype bf16 --batch 32 --tiles auto > gpurun_out/conv_bench_r02r_bf16_b32.txt 2>&1; grep -E "^#|u[248] | 56   64->  64 k3 s1| 28   96->  96 k3 s1" gpurun_out/conv_bench_r02r_bf16_b32.txt | cut -c1-120
echo "== bench f32 / bf16 bs64 / bf16 bs32"
timeout 600 python bench.py 2>gpurun_out/r_bench_default.err > gpurun_out/r_bench_default.json; cut -c1-330 gpurun_out/r_bench_default.json
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null > gpurun_out/r_bench_bf16_b64.json; cut -c1-300 gpurun_out/r_bench_bf16_b64.json
timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline 2>/dev/null > gpurun_out/r_bench_bf16_b32.json; cut -c1-300 gpurun_out/r_bench_bf16_b32.json
