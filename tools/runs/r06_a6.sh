#!/bin/bash
# GPU run 6 of round 6: same-box A/B of the input-transform arithmetic (SHAPY_W4Q_PK 0 scalar / 1 packed x pass /
# 2 packed both) x placement (SHAPY_W4Q_GROUP) of the four-wave F(4x4) kernel, end to end at bs 64.
set -u
O=gpurun_out/r06a6
mkdir -p $O
R=$PWD
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms", d["betas_sha1"])'; }
for rep in 1 2; do
  for v in pk2g1 pk0g1 pk0g12 pk1g1 pk1g12 pk2g108 pk0g108; do
    L=$R/tools/bin/libshapy_$v.so; [ $v = pk2g1 ] && L=$R/shapy_amd/csrc/libshapy_hip.so
    echo "rep $rep $v: $(SHAPY_HIP_LIB=$L bench)   unpipelined: $(SHAPY_HIP_LIB=$L bench --pipeline off)"
  done
  echo "rep $rep legacy-all: $(SHAPY_W4_LEGACY=48,96,192,384,256 bench)"
done 2>&1 | tee $O/pk_ab.txt
cls() { grep -E "^ *(56   48->  48|28   96->  96|14  192-> 192|  7  384-> 384).*r1" | cut -c1-150; }
for v in pk0g1 pk0g12 pk1g12; do
  echo "== $v" | tee -a $O/classes.txt
  SHAPY_HIP_LIB=$R/tools/bin/libshapy_$v.so timeout 200 python tools/conv_bench.py --tiles wino4,wino4k2 --iters 20 2>&1 | cls | tee -a $O/classes.txt
done
