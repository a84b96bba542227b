#!/bin/bash
# GPU run 8 of round 5: backbone latency per batch size under split-K policies (tools/ksplit_latency_sweep.py) -- is a
# small-batch policy (more slices, the 14x14 / 28x28 branches too) worth a second plan?
set -u
mkdir -p gpurun_out/r05a8
timeout 900 python tools/ksplit_latency_sweep.py --batches 1,4,8,16,32,64 --iters 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05a8/ksplit_latency_sweep.txt
