#!/bin/bash
# GPU run Y of round 4 (closing): validation of the tree (smoke, full GPU suite, the default bench line), then the
# evidence of this build: rocprofv3 kernel stats + per-phase timeline of the headline step, PMC passes
set -u
mkdir -p gpurun_out/r04y
O=$GRAFT_REPO_ROOT/gpurun_out/r04y
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/smoke.txt
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_tests_tail.txt
timeout 400 python bench.py 2>/dev/null | grep '^{' | tail -1 > $O/bench_default.json
python -c 'import json; d=json.load(open("gpurun_out/r04y/bench_default.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity"]["betas_l2"], d["cpu_baseline"]["value"]); print({k:(round(v.get("value",0),1), round(v.get("roofline",{}).get("frac",0),4), v.get("error")) for k,v in d["also"].items()})'
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-also > $O/prof_bench.json 2>$O/prof_bench.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_regressor_multistream.csv; head -8 $O/kernel_stats_regressor_multistream.csv | cut -c1-160; grep stem_conv $O/kernel_stats_regressor_multistream.csv | cut -c1-160
timeout 200 python tools/timeline.py $O/prof > $O/timeline_multistream_dag.txt 2>$O/timeline.err; tail -8 $O/timeline_multistream_dag.txt
timeout 100 python tools/phase_floors.py $O/timeline_multistream_dag.txt > $O/phase_floors.txt 2>&1; tail -9 $O/phase_floors.txt
rm -rf $O/prof
bash tools/pmc_hbm_traffic.sh gpurun_out/r04y/pmc_hbm f32 winograd4 64 2>&1 | tail -5
rm -rf $O/pmc_hbm/*/ 2>/dev/null
ls $O
