#!/bin/bash
# GPU run 8 of round 6: the merged build (old 3 + 1-wave kernel deleted, four-wave kernel = conv_wino4.hip):
# full GPU suite, smoke, PMC passes over the four-lane plan (HBM bytes per forward, MFMA busy cycles), default bench.
set -u
O=gpurun_out/r06a8
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/gpu_tests_tail.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -12 | tee $O/smoke.txt
bash tools/pmc_hbm_traffic.sh $O/pmc_hbm f32 winograd4 64 multi 2>&1 | tail -30 | tee $O/pmc_tail.txt
cp $O/pmc_hbm.json profiles/r06h_pmc_hbm_traffic_winograd4_multilane.json 2>/dev/null
timeout 600 python bench.py 2>$O/bench_stderr.txt | grep '^{' | tail -1 > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('mfma_busy'), d.get('parity',{}).get('betas_l2_mean'), d['cpu_baseline']['value'])
print({k: v for k, v in d.items() if k.startswith('also_') and k.endswith('_value')})"
