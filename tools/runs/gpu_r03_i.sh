#!/bin/bash
# ablations of the per-layer F(4x4) kernel (timing build, wrong results on purpose)
set -u
mkdir -p gpurun_out
export SHAPY_HIPCC_FLAGS='-DSHAPY_WINO_TIMING' SHAPY_HIP_LIB=/tmp/libshapy_abl.so
python -m shapy_amd.build > /dev/null 2>&1
for dbg in 0 1 2 3 4 8 12 16 32 48 7 15 31 63; do
  echo "== dbg=$dbg"
  SHAPY_WINO_DBG=$dbg timeout 200 python tools/conv_bench.py --tiles wino4 --wino4-min-hw 7 --iters 10 2>&1 | grep "wino4" | grep "r1\|256->" | cut -c1-90
done
