#!/bin/bash
# GPU run O of round 4: remaining evidence of the final build -- verbose timeline, full bench lines of the
# 256x256 crop and of configs[2]'s per-GPU shard (bf16, bs 32) with a fresh PMC traffic pass for bf16
set -u
mkdir -p gpurun_out/r04o
O=$GRAFT_REPO_ROOT/gpurun_out/r04o
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-also > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python tools/timeline.py $O/prof --verbose > $O/timeline_multistream_dag_verbose.txt 2>/dev/null; grep -c "lane" $O/timeline_multistream_dag_verbose.txt
rm -rf $O/prof
bash tools/pmc_hbm_traffic.sh gpurun_out/r04o/pmc_hbm_bf16_b32 bf16 winograd4 32 2>&1 | tail -3
cp gpurun_out/r04o/pmc_hbm_bf16_b32.json profiles/r04o_pmc_hbm_traffic_bf16_b32.json 2>/dev/null
timeout 400 python bench.py --dtype bf16 --batch 32 --no-also 2>/dev/null | grep '^{' | tail -1 > $O/bench_bf16_b32.json
timeout 400 python bench.py --size 256 --no-also 2>/dev/null | grep '^{' | tail -1 > $O/bench_f32_256.json
python -c "
import json
for f in ('bench_bf16_b32','bench_f32_256'):
    d=json.load(open('gpurun_out/r04o/%s.json'%f)); print(f, round(d['value'],1), round(d['roofline']['frac'],4), d['roofline']['traffic'], d['parity'].get('betas_l2'), d['parity'].get('features_maxabs'))
"
