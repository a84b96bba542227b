#!/bin/bash
# GPU run 9 of round 6: full GPU suite of the merged build (grouped-launch comparisons to rounding), then the
# rocprofv3 kernel-trace summary of the default bench command for profiles/.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r06a9
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/gpu_tests_tail.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-also > $O/bench_under_rocprof.out 2>/dev/null
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_regressor_pipelined.csv
head -12 $O/kernel_stats_regressor_pipelined.csv | cut -c1-150
rm -rf $O/prof
