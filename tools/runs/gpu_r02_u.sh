#!/bin/bash
# round 2, GPU run U (final): the driver's sequence (tests, smoke, bench) + the profiles DESIGN cites
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== full suite"; timeout 1200 python -m pytest tests -q -m gpu --tb=short > gpurun_out/u_all_tests.log 2>&1; tail -n 6 gpurun_out/u_all_tests.log
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
echo "== bench default (driver command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/u_bench_default.err > gpurun_out/u_bench_default.json; cut -c1-400 gpurun_out/u_bench_default.json
echo "== rocprof regressor"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/u_prof_reg -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/u_prof_reg.log 2>&1)
f=$(find gpurun_out/u_prof_reg -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/u_kernel_stats_regressor.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/u_prof_reg1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --single-stream > $R/gpurun_out/u_prof_reg1.log 2>&1)
f=$(find gpurun_out/u_prof_reg1 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/u_kernel_stats_regressor_singlestream.csv; head -14 "$f" | cut -c1-140
rm -rf gpurun_out/u_prof_reg gpurun_out/u_prof_reg1
echo "== pmc traffic f32"; timeout 900 bash tools/pmc_hbm_traffic.sh gpurun_out/u_pmc_hbm_traffic_winograd f32 winograd | grep -A8 hbm_bytes
echo "== bench measurements / smplx"; timeout 300 python bench.py --workload measurements 2>/dev/null > gpurun_out/u_bench_meas.json; cut -c1-300 gpurun_out/u_bench_meas.json
timeout 300 python bench.py --workload smplx 2>/dev/null > gpurun_out/u_bench_smplx.json; cut -c1-300 gpurun_out/u_bench_smplx.json
echo "== bf16 bs64 / bs32"
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null > gpurun_out/u_bench_bf16_b64.json; cut -c1-300 gpurun_out/u_bench_bf16_b64.json
timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline 2>/dev/null > gpurun_out/u_bench_bf16_b32.json; cut -c1-300 gpurun_out/u_bench_bf16_b32.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/u_prof_bf16 -- python $R/bench.py --dtype bf16 --batch 32 --steps 5 --warmup 3 --no-cpu-baseline --single-stream > $R/gpurun_out/u_prof_bf16.log 2>&1)
f=$(find gpurun_out/u_prof_bf16 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/u_kernel_stats_bf16_b32_singlestream.csv; rm -rf gpurun_out/u_prof_bf16
echo "== pmc traffic bf16 bs32"; timeout 600 bash tools/pmc_hbm_traffic.sh gpurun_out/u_pmc_hbm_traffic_bf16_b32 bf16 winograd 32 | grep -A8 hbm_bytes
