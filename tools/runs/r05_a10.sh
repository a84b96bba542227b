#!/bin/bash
# GPU run 10 of round 5: where configs[2]'s per-GPU shard (bf16 storage, bs 32) spends its 3.8 ms: verbose timeline
# + module tails + per-kernel stats of the bf16 forward.
set -u
mkdir -p gpurun_out/r05a10
O=$GRAFT_REPO_ROOT/gpurun_out/r05a10
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --batch 32 --steps 4 --warmup 3 --no-cpu-baseline --no-also > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python tools/timeline.py $O/prof --verbose --dtype bf16 --batch 32 > $O/timeline_bf16_b32_verbose.txt 2>$O/err.txt; tail -3 $O/err.txt
python tools/module_tails.py $O/timeline_bf16_b32_verbose.txt | tee $O/module_tails_bf16_b32.txt
grep -A12 "^phase\|total" $O/timeline_bf16_b32_verbose.txt | tail -30
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep "shapy" "$f" | cut -c1-140 | head -14
rm -rf $O/prof
