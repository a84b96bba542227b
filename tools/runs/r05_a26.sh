#!/bin/bash
# round 5, GPU run 26: the deeper cut for small batches in the product (cut at transition2 up to B = 16, workspaces with
# counters of their own): prefetch tests, product A/B at B = 1 / 8 / 16, headline bench unchanged
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a26 && export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "prefetched or next_images or two_host_threads or keep_nan" > gpurun_out/a26/tests.txt 2>&1
tail -n 3 gpurun_out/a26/tests.txt
for b in 1 8 16; do
  timeout 120 python tools/prologue_prefetch_ab.py --batch $b --steps 60 --modes base,product,base,product 2>&1 | grep -v amdgpu.ids | sed "s/^/f32 B=$b  /"
done | tee gpurun_out/a26/product_small_batches.txt
timeout 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/a26/bench.json 2> gpurun_out/a26/bench.err
python -c "
import json; d=json.loads(open('gpurun_out/a26/bench.json').read().strip().splitlines()[-1]); print('bench:', round(d['value'],1), round(d['ms_per_step'],3), d['betas_sha1'])" | tee gpurun_out/a26/bench_summary.txt
