#!/bin/bash
# round 2, GPU run L: the driver's sequence (tests, smoke, bench) + final profiles
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== full suite"; timeout 1200 python -m pytest tests -q -m gpu --tb=short > gpurun_out/l_all_tests.log 2>&1; tail -n 6 gpurun_out/l_all_tests.log
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12
echo "== bench default (driver command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/l_bench_default.err > gpurun_out/l_bench_default.json; cut -c1-400 gpurun_out/l_bench_default.json
echo "== bench direct"; timeout 600 python bench.py --algo direct --no-cpu-baseline 2>/dev/null > gpurun_out/l_bench_direct.json; cut -c1-300 gpurun_out/l_bench_direct.json
echo "== conv bench all classes"; timeout 600 python tools/conv_bench.py --tiles auto,wino > gpurun_out/conv_bench_r02l.txt 2>&1; head -2 gpurun_out/conv_bench_r02l.txt
echo "== rocprof regressor"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/l_prof_reg -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/l_prof_reg.log 2>&1)
f=$(find gpurun_out/l_prof_reg -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/l_kernel_stats_regressor.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/l_prof_reg1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --single-stream > $R/gpurun_out/l_prof_reg1.log 2>&1)
f=$(find gpurun_out/l_prof_reg1 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/l_kernel_stats_regressor_singlestream.csv
rm -rf gpurun_out/l_prof_reg gpurun_out/l_prof_reg1
echo "== pmc traffic"; timeout 900 bash tools/pmc_hbm_traffic.sh gpurun_out/l_pmc_hbm_traffic_winograd f32 winograd | grep -A8 hbm_bytes
