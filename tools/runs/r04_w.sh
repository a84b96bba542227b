#!/bin/bash
# GPU run W of round 4: is co-residency of two F(4x4) workgroups on a CU worth anything?  Timing build (wrong
# results on purpose), one vs two workgroups per CU (16 KB of unused dynamic LDS), ablation masks:
# 32 = no staging work, 8 = no filter refills, 4 = a quarter of the MFMAs, 3 = no epilogue loads / stores
set -u
mkdir -p gpurun_out/r04w
O=gpurun_out/r04w
export SHAPY_HIP_LIB=$PWD/shapy_amd/csrc/variants/libabl.so
for dyn in 0 16384; do for dbg in 0 8 32 40 44 43 47; do
  echo "== dyn_lds=$dyn dbg=$dbg"
  SHAPY_WINO_DYN_LDS=$dyn SHAPY_WINO_DBG=$dbg timeout 200 python tools/conv_bench.py --tiles wino4 --wino4-min-hw 7 --iters 10 2>&1 | grep "wino4" | grep "r1" | cut -c1-90
done; done 2>&1 | tee $O/wino4_occupancy_ablations.txt
