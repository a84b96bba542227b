#!/bin/bash
# round 2, GPU run M: no-spill Winograd kernel, float64 perimeter sum -- suite, bench, PMC traffic
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== full suite"; timeout 1200 python -m pytest tests -q -m gpu --tb=short > gpurun_out/m_all_tests.log 2>&1; tail -n 5 gpurun_out/m_all_tests.log
echo "== conv bench"; timeout 600 python tools/conv_bench.py --tiles auto,wino > gpurun_out/conv_bench_r02m.txt 2>&1; grep -E "wino|^#" gpurun_out/conv_bench_r02m.txt | cut -c1-110
echo "== bench default (driver command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/m_bench_default.err > gpurun_out/m_bench_default.json; cut -c1-330 gpurun_out/m_bench_default.json
echo "== rocprof regressor"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/m_prof_reg -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/m_prof_reg.log 2>&1)
f=$(find gpurun_out/m_prof_reg -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/m_kernel_stats_regressor.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/m_prof_reg1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --single-stream > $R/gpurun_out/m_prof_reg1.log 2>&1)
f=$(find gpurun_out/m_prof_reg1 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/m_kernel_stats_regressor_singlestream.csv
rm -rf gpurun_out/m_prof_reg gpurun_out/m_prof_reg1
echo "== pmc traffic"; timeout 900 bash tools/pmc_hbm_traffic.sh gpurun_out/m_pmc_hbm_traffic_winograd f32 winograd | grep -A8 hbm_bytes
echo "== bench measurements"; timeout 300 python bench.py --workload measurements 2>/dev/null > gpurun_out/m_bench_meas.json; cut -c1-330 gpurun_out/m_bench_meas.json
