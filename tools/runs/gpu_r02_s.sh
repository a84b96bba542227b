#!/bin/bash
# Run S: bf16 tile sweep (bs 32 and bs 64).
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 32 64; do
  timeout 500 python tools/conv_bench.py --dtype bf16 --batch $b --tiles auto,auto+bk32,64x64,64x48,128x64,128x48,64x96,128x96,64x128,128x128,256x64 > gpurun_out/conv_bench_r02s_bf16_b$b.txt 2>&1
  grep -E "^#| x *[0-9]+ \|" gpurun_out/conv_bench_r02s_bf16_b$b.txt | grep -vE " u[248] " | cut -c1-330
done
