#!/bin/bash
# round 2, GPU run A: new tests first (all failures reported), then the whole suite, the
# per-class conv bench (direct vs Winograd) and the bench workloads.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -s \
  -k "winograd or bs64 or hull_edge or 1000_meshes or lut_clamp or shipped or large_batch" > gpurun_out/a_new_tests.log 2>&1
tail -n 60 gpurun_out/a_new_tests.log
echo "== full suite"; timeout 900 python -m pytest tests -q -m gpu --tb=line > gpurun_out/a_all_tests.log 2>&1
tail -n 15 gpurun_out/a_all_tests.log
echo "== conv bench"; timeout 600 python tools/conv_bench.py --tiles auto,wino > gpurun_out/conv_bench_r02a.txt 2>&1
tail -n 50 gpurun_out/conv_bench_r02a.txt
for algo in direct auto winograd; do
  echo "== bench f32 $algo"
  timeout 300 python bench.py --algo $algo --no-cpu-baseline > gpurun_out/a_bench_f32_$algo.json 2> gpurun_out/a_bench_f32_$algo.err
  tail -c 1500 gpurun_out/a_bench_f32_$algo.json
done
echo "== bench measurements"; timeout 300 python bench.py --workload measurements > gpurun_out/a_bench_meas.json 2> gpurun_out/a_bench_meas.err; cat gpurun_out/a_bench_meas.json
echo "== bench smplx 64"; timeout 300 python bench.py --workload smplx --batch 64 > gpurun_out/a_bench_smplx64.json 2>&1; cat gpurun_out/a_bench_smplx64.json
echo "== bench smplx 4"; timeout 300 python bench.py --workload smplx --batch 4 > gpurun_out/a_bench_smplx4.json 2>&1; cat gpurun_out/a_bench_smplx4.json
echo "== bench default (with oracle)"; timeout 600 python bench.py > gpurun_out/a_bench_default.json 2> gpurun_out/a_bench_default.err; tail -c 3000 gpurun_out/a_bench_default.json
