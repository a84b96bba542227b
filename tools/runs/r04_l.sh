#!/bin/bash
# GPU run L of round 4: evidence of the current build -- rocprofv3 kernel stats + per-phase timeline of the
# headline step, PMC passes (HBM bytes, MFMA busy cycles) of one backbone forward
set -u
mkdir -p gpurun_out/r04l
O=$GRAFT_REPO_ROOT/gpurun_out/r04l
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-also > $O/prof_bench.json 2>$O/prof_bench.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_regressor_multistream.csv; head -12 $O/kernel_stats_regressor_multistream.csv | cut -c1-160
timeout 300 python tools/timeline.py $O/prof > $O/timeline_multistream_dag.txt 2>$O/timeline.err; tail -12 $O/timeline_multistream_dag.txt
timeout 100 python tools/phase_floors.py $O/timeline_multistream_dag.txt > $O/phase_floors.txt 2>&1; tail -9 $O/phase_floors.txt
rm -rf $O/prof
bash tools/pmc_hbm_traffic.sh gpurun_out/r04l/pmc_hbm f32 winograd4 64 2>&1 | tail -5
ls gpurun_out/r04l
