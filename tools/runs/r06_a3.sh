#!/bin/bash
# GPU run 3 of round 6: (a) tools/mfma_fillers.hip: what one instruction between two f32 MFMAs of a wave costs,
# by kind and placement; (b) the four-wave kernel with its staging steps in blocks of 3 / 6 / 12 instead of spread.
set -u
O=gpurun_out/r06a3
mkdir -p $O
R=$PWD
timeout 120 tools/bin/mfma_fillers 2>&1 | tee $O/mfma_fillers.txt
cls() { grep -E "^ *(56   48->  48|28   96->  96|14  192-> 192|  7  384-> 384).*r1" | cut -c1-150; }
echo "== product (group 1)" | tee $O/groups.txt
timeout 200 python tools/conv_bench.py --tiles wino4,wino4old,wino4k2 --iters 20 2>&1 | cls | tee -a $O/groups.txt
for g in 3 6 12; do
  echo "== group=$g" | tee -a $O/groups.txt
  SHAPY_HIP_LIB=$R/tools/bin/libshapy_grp$g.so timeout 200 python tools/conv_bench.py --tiles wino4,wino4k2 --iters 20 2>&1 | cls | tee -a $O/groups.txt
done
