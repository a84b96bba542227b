#!/bin/bash
set -u
for f in 7,2048,512,1 7,512,2048,1 7,2048,2048,1 7,1536,512,1 7,1536,2048,1; do
timeout 200 python tools/conv_bench.py --tiles auto,32x64,64x64,128x64,64x48,128x48,64x96,128x96,64x128,128x128 --filter $f --iters 10 2>&1 | grep -v amdgpu.ids | grep "k1"
done
