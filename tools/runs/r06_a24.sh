#!/bin/bash
# GPU run 24 of round 6: per-phase timeline / module tails / floors and the per-class conv table of the FINAL build.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r06a24
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-also --pipeline off > $O/bench_under_rocprof.out 2>/dev/null
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_regressor_multistream.csv
timeout 300 python tools/timeline.py $O/prof > $O/timeline.txt 2>$O/err.txt; tail -8 $O/timeline.txt
timeout 300 python tools/timeline.py $O/prof --verbose > $O/timeline_verbose.txt 2>>$O/err.txt
timeout 300 python tools/module_tails.py $O/timeline_verbose.txt > $O/module_tails.txt 2>>$O/err.txt; cat $O/module_tails.txt
timeout 300 python tools/phase_floors.py --timeline $O/timeline.txt > $O/phase_floors.txt 2>>$O/err.txt; cat $O/phase_floors.txt
rm -rf $O/prof
timeout 600 python tools/conv_bench.py --tiles auto,wino,wino4,wino4k2 --iters 20 > $O/conv_bench_all_classes.txt 2>&1; tail -60 $O/conv_bench_all_classes.txt | cut -c1-170
