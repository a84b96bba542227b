#!/bin/bash
# Run T: bf16 without the 48 -> 64 channel padding (flat-K kernel).
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bf16 tests"
timeout 900 python -m pytest tests -q -m gpu -x -k "bf16" 2>&1 | tail -6
echo "== bench bf16 bs64 / bs32"
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline 2>gpurun_out/t_bench_bf16.err > gpurun_out/t_bench_bf16_b64.json; cut -c1-300 gpurun_out/t_bench_bf16_b64.json; tail -3 gpurun_out/t_bench_bf16.err
timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline 2>/dev/null > gpurun_out/t_bench_bf16_b32.json; cut -c1-300 gpurun_out/t_bench_bf16_b32.json
echo "== conv bench bf16 bs64"
timeout 400 python tools/conv_bench.py --dtype bf16 --batch 64 --tiles auto > gpurun_out/conv_bench_r02t_bf16_b64.txt 2>&1; grep -E "^#| 48->|->  48 " gpurun_out/conv_bench_r02t_bf16_b64.txt | cut -c1-120
