#!/bin/bash
# round 2, GPU run G: hull v6 (64-bit-key rank sort, (x,z) dedupe) -- tests, timing, benches, stats
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== tests"; timeout 1200 python -m pytest tests -q -m gpu --tb=short > gpurun_out/g_all_tests.log 2>&1; tail -n 12 gpurun_out/g_all_tests.log
echo "== measurement phase timing"; timeout 600 python tools/measure_timing.py 2>&1 | tail -6
echo "== bench measurements"; timeout 300 python bench.py --workload measurements 2>/dev/null > gpurun_out/g_bench_meas.json; cut -c1-420 gpurun_out/g_bench_meas.json
echo "== rocprof measurements"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g_prof_meas -- python $R/bench.py --workload measurements --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/g_prof_meas.log 2>&1)
f=$(find gpurun_out/g_prof_meas -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/g_kernel_stats_measurements.csv; head -3 "$f" | cut -c1-50,150-260
rm -rf gpurun_out/g_prof_meas
echo "== bench default"; timeout 600 python bench.py 2> gpurun_out/g_bench_default.err > gpurun_out/g_bench_default.json; cut -c1-330 gpurun_out/g_bench_default.json
echo "== rocprof regressor"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g_prof_reg -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/g_prof_reg.log 2>&1)
f=$(find gpurun_out/g_prof_reg -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/g_kernel_stats_regressor.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g_prof_reg1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --single-stream > $R/gpurun_out/g_prof_reg1.log 2>&1)
f=$(find gpurun_out/g_prof_reg1 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/g_kernel_stats_regressor_singlestream.csv
rm -rf gpurun_out/g_prof_reg gpurun_out/g_prof_reg1
grep -v "conv_\|rocclr_copy\|direct_copy" gpurun_out/g_kernel_stats_regressor.csv | cut -c1-70,120-200 | head -14
