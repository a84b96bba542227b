#!/bin/bash
# round 2, GPU run E: measurement phase timing, N-slab tile order A/B, quick tests, bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== measurement phase timing"; timeout 600 python tools/measure_timing.py 2>&1 | tail -12
echo "== tests (conv + hrnet + bs64)"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -k "conv or hrnet or bs64" > gpurun_out/e_tests.log 2>&1; tail -n 8 gpurun_out/e_tests.log
echo "== conv bench head layers: N-slab vs m-major"; timeout 600 python tools/conv_bench.py --tiles auto,auto+nonslab,wino > gpurun_out/conv_bench_r02e.txt 2>&1
grep -E "^#|^  7 " gpurun_out/conv_bench_r02e.txt
echo "== bench default"; timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/e_bench_default.err > gpurun_out/e_bench_default.json; cut -c1-330 gpurun_out/e_bench_default.json
