#!/bin/bash
# round 2, GPU run C: full suite on the fused tail, measurement ablations, kernel stats, PMC calibration
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== full suite"; timeout 1200 python -m pytest tests -q -m gpu --tb=short > gpurun_out/c_all_tests.log 2>&1
tail -n 40 gpurun_out/c_all_tests.log
echo "== bench measurements"; timeout 300 python bench.py --workload measurements 2>/dev/null > gpurun_out/c_bench_meas.json; cut -c1-700 gpurun_out/c_bench_meas.json
for dbg in 1 3 7; do
  echo "== measurements ablation dbg=$dbg"
  SHAPY_MEASURE_DBG=$dbg timeout 300 python bench.py --workload measurements --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done
echo "== rocprof measurements"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c_prof_meas -- python $R/bench.py --workload measurements --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/c_prof_meas.log 2>&1)
f=$(find gpurun_out/c_prof_meas -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/c_kernel_stats_measurements.csv; head -4 "$f" | cut -c1-160
echo "== bench default (with oracle)"; timeout 600 python bench.py 2> gpurun_out/c_bench_default.err > gpurun_out/c_bench_default.json; cut -c1-1200 gpurun_out/c_bench_default.json
echo "== rocprof regressor"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c_prof_reg -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/c_prof_reg.log 2>&1)
f=$(find gpurun_out/c_prof_reg -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/c_kernel_stats_regressor.csv; grep -v "conv_" "$f" | head -30 | cut -c1-200
echo "== bench smplx"; timeout 300 python bench.py --workload smplx --batch 64 2>/dev/null | cut -c1-900
echo "== pmc calibration"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/c_pmc_cal -- python $R/tools/pmc_calibrate.py > $R/gpurun_out/c_pmc_cal_fetch.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/c_pmc_cal -- python $R/tools/pmc_calibrate.py > $R/gpurun_out/c_pmc_cal_write.log 2>&1)
python tools/pmc_calibrate.py --parse gpurun_out/c_pmc_cal > gpurun_out/c_pmc_calibration.json; cat gpurun_out/c_pmc_calibration.json
rm -rf gpurun_out/c_prof_meas gpurun_out/c_prof_reg gpurun_out/c_pmc_cal
