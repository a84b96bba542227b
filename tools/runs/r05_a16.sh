#!/bin/bash
# round 5, GPU run 16: next batch's stem + layer1 under the current batch (prototype A/B) + network tests after the
# transform memo
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a16 && export TMPDIR=/tmp
timeout 600 python tools/prologue_prefetch_ab.py --batch 64 > gpurun_out/a16/prefetch_f32_b64.txt 2>&1
timeout 300 python tools/prologue_prefetch_ab.py --batch 32 --dtype bf16 --steps 60 > gpurun_out/a16/prefetch_bf16_b32.txt 2>&1
timeout 300 python tools/prologue_prefetch_ab.py --batch 8 --steps 60 --modes base,split,before-3,after-3,base > gpurun_out/a16/prefetch_f32_b8.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "hrnet or full_forward" > gpurun_out/a16/tests_network.txt 2>&1
tail -n 12 gpurun_out/a16/*.txt
