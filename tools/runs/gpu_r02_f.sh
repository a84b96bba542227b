#!/bin/bash
# round 2, GPU run F: hull v5 (rank sort + elimination rounds), specialised candidate SAT, N-slab traffic
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== tests"; timeout 1200 python -m pytest tests -q -m gpu --tb=short > gpurun_out/f_all_tests.log 2>&1; tail -n 12 gpurun_out/f_all_tests.log
echo "== measurement phase timing"; timeout 600 python tools/measure_timing.py 2>&1 | tail -6
echo "== bench measurements"; timeout 300 python bench.py --workload measurements 2>/dev/null > gpurun_out/f_bench_meas.json; cut -c1-420 gpurun_out/f_bench_meas.json
echo "== rocprof measurements"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f_prof_meas -- python $R/bench.py --workload measurements --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/f_prof_meas.log 2>&1)
f=$(find gpurun_out/f_prof_meas -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/f_kernel_stats_measurements.csv; head -3 "$f" | cut -c1-50,150-260
rm -rf gpurun_out/f_prof_meas
echo "== bench default"; timeout 600 python bench.py 2> gpurun_out/f_bench_default.err > gpurun_out/f_bench_default.json; cut -c1-330 gpurun_out/f_bench_default.json
echo "== pmc traffic (winograd + N-slab)"; timeout 900 bash tools/pmc_hbm_traffic.sh gpurun_out/f_pmc_hbm_traffic_winograd f32 winograd | grep -A8 hbm_bytes
