#!/bin/bash
# round 5, GPU run 17: forward(x, prefetch=next_x) in the product: tests, A/B through the product API, bench on / off
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a17 && export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "prefetched or next_images or two_host_threads or event_driven" > gpurun_out/a17/tests_prefetch.txt 2>&1
timeout 300 python tools/prologue_prefetch_ab.py --batch 64 --modes base,product,after-3,base,product > gpurun_out/a17/prefetch_product_f32_b64.txt 2>&1
( time timeout 400 python bench.py ) > gpurun_out/a17/bench_default.json 2> gpurun_out/a17/bench_default.err
timeout 300 python bench.py --pipeline off --no-also --no-cpu-baseline > gpurun_out/a17/bench_pipeline_off.json 2> gpurun_out/a17/bench_pipeline_off.err
tail -n 8 gpurun_out/a17/tests_prefetch.txt gpurun_out/a17/prefetch_product_f32_b64.txt gpurun_out/a17/bench_default.err
python - <<'PY'
import json
for f in ('bench_default', 'bench_pipeline_off'):
    try:
        r = json.loads(open(f'gpurun_out/a17/{f}.json').read().strip().splitlines()[-1])
        print(f, r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline'].get('duration'), r['config'].get('pipelined_batches'))
        print({k: v for k, v in r.items() if k.startswith('also_') and k.endswith('_value')})
        print(r.get('parity'))
    except Exception as e:
        print(f, 'ERR', e)
PY
