#!/bin/bash
# GPU run 13 of round 6: timing builds of the four-wave kernel in STEADY STATE (B = 256: three rounds of two
# workgroups per CU) -- SHAPY_W4Q_DBG masks: 1 transform VALU, 2 patch loads, 4 V writes, 8 filter refills, 16 V reads.
set -u
O=gpurun_out/r06a13
mkdir -p $O
R=$PWD
cls() { grep -E "^ *(56   48->  48|28   96->  96|14  192-> 192|  7  384-> 384).*r1" | cut -c1-150; }
for d in 0 1 2 7 8 15 16 31; do
  L=$R/tools/bin/libshapy_dbg$d.so; [ $d = 0 ] && L=$R/shapy_amd/csrc/libshapy_hip.so
  echo "== dbg=$d" | tee -a $O/ablations_b256.txt
  SHAPY_HIP_LIB=$L timeout 300 python tools/conv_bench.py --tiles wino4 --iters 10 --batch 256 2>&1 | cls | tee -a $O/ablations_b256.txt
done
