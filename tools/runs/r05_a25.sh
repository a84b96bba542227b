#!/bin/bash
# round 5, GPU run 25: deeper cuts of the batch pipeline at SMALL (latency-bound) batches
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a25 && export TMPDIR=/tmp
{
for b in 1 8; do
  for bar in 1 2 3; do
    timeout 120 python tools/prologue_prefetch_ab.py --batch $b --steps 60 --barrier $bar --modes base,before-3,after-3 2>&1 | grep -v amdgpu.ids | sed "s/^/f32 B=$b barrier $bar  /"
  done
done
for bar in 2 3; do
  timeout 120 python tools/prologue_prefetch_ab.py --batch 16 --steps 60 --barrier $bar --modes base,before-3,after-3 2>&1 | grep -v amdgpu.ids | sed "s/^/f32 B=16 barrier $bar  /"
done
timeout 120 python tools/prologue_prefetch_ab.py --batch 32 --dtype bf16 --steps 60 --barrier 3 --modes base,before-3,after-3 2>&1 | grep -v amdgpu.ids | sed "s/^/bf16 B=32 barrier 3  /"
} > gpurun_out/a25/prefetch_deeper_cuts_small_batches.txt
cat gpurun_out/a25/prefetch_deeper_cuts_small_batches.txt
