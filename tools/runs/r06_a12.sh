#!/bin/bash
# GPU run 12 of round 6: steady-state rate of the four-wave F(4x4) kernel (many rounds of workgroups: B = 256 / 512),
# to separate what the kernel loses by itself from what a 1.5-round launch at B = 64 loses.
set -u
O=gpurun_out/r06a12
mkdir -p $O
cls() { grep -E "^ *(56   48->  48|28   96->  96|14  192-> 192|  7  384-> 384).*r1" | cut -c1-150; }
for b in 64 128 256 512; do
  echo "== B=$b" | tee -a $O/steady.txt
  timeout 300 python tools/conv_bench.py --tiles wino4,wino4k2 --iters 10 --batch $b 2>&1 | cls | tee -a $O/steady.txt
done
