#!/bin/bash
# GPU run 4 of round 6: end-to-end A/B at bs 64 -- which layer classes gain from the four-wave F(4x4) kernel
# inside the four-lane forward (co-resident workgroups of other branches) as opposed to isolated launches.
set -u
O=gpurun_out/r06a4
mkdir -p $O
R=$PWD
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms", d["betas_sha1"])'; }
for rep in 1 2; do
  for leg in "" "384" "192,384" "256,192,384" "96,192,384" "48,96,192,384,256"; do
    echo "rep $rep legacy='$leg': $(SHAPY_W4_LEGACY="$leg" bench)  unpipelined: $(SHAPY_W4_LEGACY="$leg" bench --pipeline off)"
  done
  echo "rep $rep group12 all-new: $(SHAPY_HIP_LIB=$R/tools/bin/libshapy_grp12.so bench)"
done 2>&1 | tee $O/legacy_ab.txt
