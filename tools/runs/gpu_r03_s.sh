#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s_gpu_tests.log 2>&1; tail -6 gpurun_out/s_gpu_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/s_bench.json 2>gpurun_out/s_bench.err; cut -c1-1200 gpurun_out/s_bench.json; tail -2 gpurun_out/s_bench.err
