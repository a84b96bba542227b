#!/bin/bash
# GPU run 15 of round 6: configs[2] (bf16, bs 32 per GPU): eager four lanes against hipGraph replay / one stream --
# is the shard bound by launch dispatch?
set -u
O=gpurun_out/r06a15
mkdir -p $O
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also --dtype bf16 --batch 32 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", round(d["ms_per_step"],3), "ms per step", d["config"].get("hip_graph"))'; }
for rep in 1 2; do
  echo "rep $rep eager 4 lanes pipelined: $(bench)"
  echo "rep $rep eager 4 lanes one at a time: $(bench --pipeline off)"
  echo "rep $rep graph on: $(bench --graph on --pipeline off)"
  echo "rep $rep single stream: $(bench --single-stream --pipeline off)"
  echo "rep $rep single stream graph on: $(bench --single-stream --graph on --pipeline off)"
done 2>&1 | tee $O/bf16_b32_modes.txt
