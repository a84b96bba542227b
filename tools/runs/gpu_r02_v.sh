#!/bin/bash
# round 2, GPU run V: last build -- suite, smoke, bench (D2H of the betas inside the timed region)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== full suite"; timeout 1200 python -m pytest tests -q -m gpu --tb=short > gpurun_out/v_all_tests.log 2>&1; tail -n 4 gpurun_out/v_all_tests.log
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench default (driver command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/v_bench_default.err > gpurun_out/v_bench_default.json; cut -c1-330 gpurun_out/v_bench_default.json
echo "== bench bf16 bs32"; timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline 2>/dev/null > gpurun_out/v_bench_bf16_b32.json; cut -c90-330 gpurun_out/v_bench_bf16_b32.json
