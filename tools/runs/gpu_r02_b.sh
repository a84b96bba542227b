#!/bin/bash
# round 2, GPU run B: Winograd TM variants, barrier-free measurement scan, kernel stats, PMC traffic
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== tests"; timeout 900 python -m pytest tests -q -m gpu --tb=short \
  -k "winograd or bs64 or hull_edge or 1000_meshes or shipped or large_batch or evaluate_hbw or virtual_measurements" > gpurun_out/b_tests.log 2>&1
tail -n 40 gpurun_out/b_tests.log
echo "== conv bench"; timeout 600 python tools/conv_bench.py --tiles auto,wino1,wino2 > gpurun_out/conv_bench_r02b.txt 2>&1
grep -E "wino|^#" gpurun_out/conv_bench_r02b.txt
echo "== bench measurements"; timeout 300 python bench.py --workload measurements 2>/dev/null > gpurun_out/b_bench_meas.json; cat gpurun_out/b_bench_meas.json | cut -c1-900
echo "== bench default (winograd, with oracle)"; timeout 600 python bench.py 2> gpurun_out/b_bench_default.err > gpurun_out/b_bench_default.json; cut -c1-1500 gpurun_out/b_bench_default.json
echo "== rocprof measurements"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b_prof_meas -- python $R/bench.py --workload measurements --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/b_prof_meas.log 2>&1)
f=$(find gpurun_out/b_prof_meas -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
echo "== rocprof regressor"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b_prof_reg -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/b_prof_reg.log 2>&1)
f=$(find gpurun_out/b_prof_reg -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/b_kernel_stats_regressor.csv; head -30 "$f" | cut -c1-220
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b_prof_reg1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --single-stream > $R/gpurun_out/b_prof_reg1.log 2>&1)
f=$(find gpurun_out/b_prof_reg1 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/b_kernel_stats_regressor_singlestream.csv
f=$(find gpurun_out/b_prof_meas -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/b_kernel_stats_measurements.csv
rm -rf gpurun_out/b_prof_meas gpurun_out/b_prof_reg gpurun_out/b_prof_reg1
echo "== pmc traffic (winograd)"; timeout 900 bash tools/pmc_hbm_traffic.sh gpurun_out/b_pmc_hbm_traffic_winograd f32 winograd | tail -30
