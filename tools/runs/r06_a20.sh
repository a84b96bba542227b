#!/bin/bash
# GPU run 20 of round 6: order / grouping of the staging micro-steps of the F(4x4) kernel (patch loads last), same-box A/B.
set -u
O=gpurun_out/r06a20
mkdir -p $O
R=$PWD
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms", d["betas_sha1"])'; }
for rep in 1 2; do
  for v in product ll1g12 ll1g6 ll0g6; do
    L=$R/tools/bin/libshapy_$v.so; [ $v = product ] && L=$R/shapy_amd/csrc/libshapy_hip.so
    echo "rep $rep $v: $(SHAPY_HIP_LIB=$L bench)   one at a time: $(SHAPY_HIP_LIB=$L bench --pipeline off)"
  done
done 2>&1 | tee $O/loads_last_ab.txt
cls() { grep -E "^ *(56   48->  48|28   96->  96|14  192-> 192|  7  384-> 384).*r1" | cut -c1-150; }
for v in product ll1g12; do
  L=$R/tools/bin/libshapy_$v.so; [ $v = product ] && L=$R/shapy_amd/csrc/libshapy_hip.so
  echo "== $v B=256" | tee -a $O/classes.txt
  SHAPY_HIP_LIB=$L timeout 200 python tools/conv_bench.py --tiles wino4 --iters 10 --batch 256 2>&1 | cls | tee -a $O/classes.txt
done
