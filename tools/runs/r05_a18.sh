#!/bin/bash
# round 5, GPU run 18: a deeper cut (stem + layer1 + stage 2 as the prefetched prologue)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a18 && export TMPDIR=/tmp
timeout 400 python tools/prologue_prefetch_ab.py --batch 64 --barrier 2 --modes base,split,before-3,after-3,before-2,after-2,base > gpurun_out/a18/prefetch_cut2_f32_b64.txt 2>&1
timeout 300 python tools/prologue_prefetch_ab.py --batch 32 --dtype bf16 --steps 60 --barrier 2 --modes base,before-3,after-3,base > gpurun_out/a18/prefetch_cut2_bf16_b32.txt 2>&1
timeout 300 python tools/prologue_prefetch_ab.py --batch 64 --barrier 3 --modes base,before-3,after-3,base > gpurun_out/a18/prefetch_cut3_f32_b64.txt 2>&1
tail -n 9 gpurun_out/a18/*.txt
