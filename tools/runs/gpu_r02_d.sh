#!/bin/bash
# round 2, GPU run D: hull v4 (register sort), trimmed scan loop, NN=4 Winograd, per-launch PMC traffic
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== tests"; timeout 1200 python -m pytest tests -q -m gpu --tb=short > gpurun_out/d_all_tests.log 2>&1
tail -n 30 gpurun_out/d_all_tests.log
echo "== bench measurements"; timeout 300 python bench.py --workload measurements 2>/dev/null > gpurun_out/d_bench_meas.json; cut -c1-600 gpurun_out/d_bench_meas.json
for dbg in 2 1; do
  echo "== measurements ablation dbg=$dbg"
  SHAPY_MEASURE_DBG=$dbg timeout 300 python bench.py --workload measurements --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done
echo "== rocprof measurements"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/d_prof_meas -- python $R/bench.py --workload measurements --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/d_prof_meas.log 2>&1)
f=$(find gpurun_out/d_prof_meas -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/d_kernel_stats_measurements.csv; head -3 "$f" | cut -c1-60,150-260
echo "== conv bench (winograd classes)"; timeout 600 python tools/conv_bench.py --tiles auto,wino > gpurun_out/conv_bench_r02d.txt 2>&1
grep -E "wino|^#" gpurun_out/conv_bench_r02d.txt
echo "== bench default (with oracle)"; timeout 600 python bench.py 2> gpurun_out/d_bench_default.err > gpurun_out/d_bench_default.json; cut -c1-400 gpurun_out/d_bench_default.json
echo "== rocprof regressor"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/d_prof_reg -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/d_prof_reg.log 2>&1)
f=$(find gpurun_out/d_prof_reg -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/d_kernel_stats_regressor.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/d_prof_reg1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --single-stream > $R/gpurun_out/d_prof_reg1.log 2>&1)
f=$(find gpurun_out/d_prof_reg1 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/d_kernel_stats_regressor_singlestream.csv
rm -rf gpurun_out/d_prof_meas gpurun_out/d_prof_reg gpurun_out/d_prof_reg1
echo "== pmc traffic (winograd)"; timeout 900 bash tools/pmc_hbm_traffic.sh gpurun_out/d_pmc_hbm_traffic_winograd f32 winograd | tail -22
head -5 gpurun_out/d_pmc_hbm_traffic_winograd_per_launch.csv
