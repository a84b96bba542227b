#!/bin/bash
# Run Q: three chunks of global loads in flight in the implicit-GEMM kernel (bf16 default, f32 A/B).
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bf16 + conv tests"
timeout 600 python -m pytest tests -q -m gpu -x -k "bf16 or conv_kernel" 2>&1 | tail -4
echo "== conv bench bf16 bs32: pd3 (auto) vs pd1"
timeout 400 python tools/conv_bench.py --dtype bf16 --batch 32 --tiles auto,auto+pd1 > gpurun_out/conv_bench_r02q_bf16_b32.txt 2>&1; cut -c1-150 gpurun_out/conv_bench_r02q_bf16_b32.txt | tail -50
echo "== conv bench f32 bs64: auto vs pd3"
timeout 400 python tools/conv_bench.py --tiles auto,auto+pd3 > gpurun_out/conv_bench_r02q_f32_pd3.txt 2>&1; cut -c1-150 gpurun_out/conv_bench_r02q_f32_pd3.txt | tail -50
echo "== bench bf16 bs64 / bs32"
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null > gpurun_out/q_bench_bf16_b64.json; cut -c1-300 gpurun_out/q_bench_bf16_b64.json
timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline 2>/dev/null > gpurun_out/q_bench_bf16_b32.json; cut -c1-300 gpurun_out/q_bench_bf16_b32.json
