#!/bin/bash
# round 2, GPU run X (the round's last GPU minutes): the driver's sequence under the NEW default
# (conv_algo = 'winograd4', F(4x4) from 14 px) -- whole suite, bench, kernel stats
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S0=$SECONDS
stamp() { echo "[t=$((SECONDS - S0))s] $*"; }
stamp "full suite"
timeout 400 python -m pytest tests -q -m gpu --tb=short > gpurun_out/x_all_tests.log 2>&1
tail -n 5 gpurun_out/x_all_tests.log
stamp "driver bench"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/x_bench_default.err \
    > gpurun_out/x_bench_default.json
cut -c1-330 gpurun_out/x_bench_default.json
stamp "rocprof kernel stats, single stream"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/x_prof1 \
    -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --single-stream \
    > $R/gpurun_out/x_prof1.log 2>&1)
f=$(find gpurun_out/x_prof1 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/x_kernel_stats_regressor_singlestream.csv && head -9 "$f" | cut -c1-140
rm -rf gpurun_out/x_prof1
stamp "rocprof kernel stats, multi stream"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/x_prof \
    -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/x_prof.log 2>&1)
f=$(find gpurun_out/x_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/x_kernel_stats_regressor_multistream.csv
rm -rf gpurun_out/x_prof
stamp "smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
stamp "done"
