#!/bin/bash
# GPU run 26 of round 6: the K-loop skeleton with filter rings of 3 / 9 / 27 items: how much of the f32 K loop's loss is
# the ring's lead?
set -u
mkdir -p gpurun_out/r06a26
for r in 3 9 27; do
  b=tools/bin/w4_bf16x3_skeleton_r$r; [ $r = 9 ] && b=tools/bin/w4_bf16x3_skeleton
  echo "== ring $r"; timeout 120 $b 2>&1 | awk 'NR%3==0'
done | tee gpurun_out/r06a26/skeleton_ring_depth.txt
