#!/bin/bash
# round 5, GPU run 20: the head's wide 1x1 GEMMs on the bf16x6 kernel inside the float32 plan (SHAPY_TILE_X6): tests + bench A/B
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a20 && export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "f32x6 or head_gemms or bs64 or prefetched_prologue or event_driven" > gpurun_out/a20/tests_x6.txt 2>&1
timeout 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/a20/bench_x6_head.json 2> gpurun_out/a20/bench_x6_head.err
SHAPY_X6_GEMM_MIN_BATCH=0 timeout 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/a20/bench_f32_head.json 2> gpurun_out/a20/bench_f32_head.err
timeout 300 python bench.py --no-also > gpurun_out/a20/bench_x6_head_parity.json 2> gpurun_out/a20/bench_x6_head_parity.err
tail -n 6 gpurun_out/a20/tests_x6.txt
python - <<'PY'
import json
for f in ('bench_x6_head', 'bench_f32_head', 'bench_x6_head_parity'):
    try:
        r = json.loads(open(f'gpurun_out/a20/{f}.json').read().strip().splitlines()[-1])
        print(f, round(r['value'], 1), round(r['ms_per_step'], 3), round(r['roofline']['frac'], 4), r['config'].get('head_gemm_arithmetic'), r['betas_sha1'])
        if 'parity' in r: print(r['parity'])
    except Exception as e:
        print(f, 'ERR', e); print(open(f'gpurun_out/a20/{f}.err').read()[-1500:])
PY
