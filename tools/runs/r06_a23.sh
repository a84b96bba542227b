#!/bin/bash
# GPU run 23 of round 6: the FINAL tree (A/B knobs of the round removed): full GPU suite, smoke, default bench.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r06a23
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $O/gpu_tests_tail.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -11 | tee $O/smoke.txt
( time timeout 900 python bench.py 2>$O/bench_stderr.txt | grep '^{' | tail -1 > $O/bench_default.json ) 2>&1 | grep real | tee $O/bench_wall.txt
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac']); print('mfma_busy', d['roofline'].get('mfma_busy')); print('parity betas_l2', d['parity']['betas_l2'], 'vertices', d['parity']['vertices_maxabs']); print('cpu', d['cpu_baseline']['value'])
print({k: round(v, 1) for k, v in d.items() if k.startswith('also_') and k.endswith('_value')})" | tee $O/bench_summary.txt
