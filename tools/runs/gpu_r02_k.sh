#!/bin/bash
# round 2, GPU run K: B-fragment refills pinned after their MFMA group
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests (winograd + bs64)"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -k "winograd or bs64" > gpurun_out/k_tests.log 2>&1; tail -n 4 gpurun_out/k_tests.log
echo "== winograd phase timing"; timeout 900 python tools/wino_timing.py 2>&1 | tail -6
echo "== conv bench"; timeout 600 python tools/conv_bench.py --tiles wino,wino1,wino2 --iters 12 > gpurun_out/conv_bench_r02k.txt 2>&1; grep -E "wino|^#" gpurun_out/conv_bench_r02k.txt | cut -c1-130
echo "== bench default"; timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/k_bench_default.err > gpurun_out/k_bench_default.json; cut -c1-330 gpurun_out/k_bench_default.json
