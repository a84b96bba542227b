#!/bin/bash
# GPU run 15 of round 5 (closing evidence): rocprofv3 kernel stats + per-phase timeline + floors of the default
# configuration, PMC passes (HBM bytes per forward, MFMA busy cycles) of the closing build.
set -u
mkdir -p gpurun_out/r05a15
O=$GRAFT_REPO_ROOT/gpurun_out/r05a15
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-also > $O/bench_under_rocprof.out 2>/dev/null
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_regressor_multistream.csv
head -8 $O/kernel_stats_regressor_multistream.csv | cut -c1-140
timeout 300 python tools/timeline.py $O/prof > $O/timeline_multistream_dag.txt 2>$O/timeline_err.txt; tail -12 $O/timeline_multistream_dag.txt
timeout 300 python tools/phase_floors.py $O/timeline_multistream_dag.txt > $O/phase_floors.txt 2>>$O/timeline_err.txt; cat $O/phase_floors.txt
rm -rf $O/prof
bash tools/pmc_hbm_traffic.sh gpurun_out/r05a15/pmc_hbm f32 winograd4 64 2>&1 | tail -4
ls gpurun_out/r05a15/
