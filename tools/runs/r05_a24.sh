#!/bin/bash
# round 5, GPU run 24: the N-rank step rehearsed on one GPU (--force-gather: ncclAllGather on lane 1's stream) with the
# batch pipeline on lane 3's stream, on / off
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a24 && export TMPDIR=/tmp
for p in on off; do
  timeout 300 python bench.py --force-gather --pipeline $p --no-also --no-cpu-baseline > gpurun_out/a24/bench_force_gather_pipeline_$p.json 2> gpurun_out/a24/err_$p.txt
  python - <<PY
import json
try:
    r = json.loads(open('gpurun_out/a24/bench_force_gather_pipeline_$p.json').read().strip().splitlines()[-1])
    print('pipeline $p:', round(r['value'], 1), 'img/s', round(r['ms_per_step'], 3), 'ms', r.get('force_gather'), r['betas_sha1'])
except Exception as e:
    print('pipeline $p ERR', e); print(open('gpurun_out/a24/err_$p.txt').read()[-2000:])
PY
done | tee gpurun_out/a24/summary.txt
