#!/bin/bash
# round 2, GPU run I: residual prefetch in the last Winograd chunk
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests (winograd + hrnet + bs64)"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -k "winograd or hrnet or bs64" > gpurun_out/i_tests.log 2>&1; tail -n 6 gpurun_out/i_tests.log
echo "== winograd phase timing"; timeout 900 python tools/wino_timing.py 2>&1 | tail -6
echo "== conv bench"; timeout 600 python tools/conv_bench.py --tiles auto,wino > gpurun_out/conv_bench_r02i.txt 2>&1; grep -E "wino|^#" gpurun_out/conv_bench_r02i.txt
echo "== bench default"; timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/i_bench_default.err > gpurun_out/i_bench_default.json; cut -c1-330 gpurun_out/i_bench_default.json
