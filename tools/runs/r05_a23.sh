#!/bin/bash
# round 5, GPU run 23: where the issue order of the prefetched prologue flips (float32, B = 16 / 32; bf16 B = 8 / 64)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a23 && export TMPDIR=/tmp
for b in 16 32; do
  timeout 200 python tools/prologue_prefetch_ab.py --batch $b --steps 40 --modes base,before-3,after-3,base 2>&1 | grep -v amdgpu.ids | sed "s/^/f32 B=$b  /"
done > gpurun_out/a23/prefetch_order_by_batch.txt
for b in 8 64; do
  timeout 200 python tools/prologue_prefetch_ab.py --batch $b --dtype bf16 --steps 40 --modes base,before-3,after-3,base 2>&1 | grep -v amdgpu.ids | sed "s/^/bf16 B=$b  /"
done >> gpurun_out/a23/prefetch_order_by_batch.txt
cat gpurun_out/a23/prefetch_order_by_batch.txt
