#!/bin/bash
# GPU run 7 of round 6: (a) lane stream priorities re-tuned for the four-wave kernel (lane 0 = the 56x56 branch ends a
# stage-3 / stage-4 module last); (b) kernel trace + per-phase timeline + module tails of the new default build.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r06a7
mkdir -p $O
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>>$O/stderr.txt | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms", d["betas_sha1"])'; }
for rep in 1 2; do
  for pr in "1,2,3" "0,0,0" "1,1,1" "0,1,2" "3,2,1" "0,0,1" "2,2,2"; do
    echo "rep $rep prio=$pr: $(SHAPY_LANE_PRIO=$pr bench)   unpipelined: $(SHAPY_LANE_PRIO=$pr bench --pipeline off)"
  done
done 2>&1 | tee $O/lane_prio.txt
grep "priority range" $O/stderr.txt | head -1 | tee -a $O/lane_prio.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-also --pipeline off > $O/bench_under_rocprof.out 2>/dev/null
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_regressor_multistream.csv
timeout 300 python tools/timeline.py $O/prof > $O/timeline.txt 2>$O/timeline_err.txt; tail -25 $O/timeline.txt
timeout 300 python tools/timeline.py $O/prof --verbose > $O/timeline_verbose.txt 2>>$O/timeline_err.txt
timeout 300 python tools/module_tails.py $O/timeline_verbose.txt > $O/module_tails.txt 2>>$O/timeline_err.txt; cat $O/module_tails.txt
timeout 300 python tools/phase_floors.py $O/timeline.txt > $O/phase_floors.txt 2>>$O/timeline_err.txt; cat $O/phase_floors.txt
rm -rf $O/prof
