#!/bin/bash
# GPU run 18 of round 6: order of the terms of a fuse output (nearest stride-2 branch first against ascending): A/B at
# bs 64 (+ bf16 bs 32), parity of the new order.
set -u
O=gpurun_out/r06a18
mkdir -p $O
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms", d["betas_sha1"])'; }
for rep in 1 2 3; do
  echo "rep $rep near-first: $(bench)   one at a time: $(bench --pipeline off)"
  echo "rep $rep ascending: $(SHAPY_FUSE_NEAR_FIRST=0 bench)   one at a time: $(SHAPY_FUSE_NEAR_FIRST=0 bench --pipeline off)"
done 2>&1 | tee $O/fuse_order_ab.txt
echo "bf16 b32 near-first: $(bench --dtype bf16 --batch 32)  ascending: $(SHAPY_FUSE_NEAR_FIRST=0 bench --dtype bf16 --batch 32)" | tee -a $O/fuse_order_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_features or hrnet_features_vs_reference_golden or full_forward_bs64 or event_driven_plan_equals or features_256 or full_forward_vs_reference" 2>&1 | tail -4 | tee $O/tests.txt
