#!/bin/bash
# GPU run 16 of round 6: bf16 branch levels as one flat-K launch (conv_igemm_group_kernel): parity, then configs[2]'s
# shard (bs 32) and bs 64 with and without.
set -u
O=gpurun_out/r06a16
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "group_bf16 or bf16_grouped or bf16_features or kernel_bf16 or conv2d_group_c_abi" 2>&1 | tail -6 | tee $O/tests.txt
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also --dtype bf16 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", round(d["ms_per_step"],3), "ms per step")'; }
for rep in 1 2; do
  for b in 32 64; do
    echo "rep $rep B=$b grouped: $(bench --batch $b)   one at a time: $(bench --batch $b --pipeline off)"
    echo "rep $rep B=$b per-layer: $(SHAPY_GROUP_BRANCHES_BF16=0 bench --batch $b)   one at a time: $(SHAPY_GROUP_BRANCHES_BF16=0 bench --batch $b --pipeline off)"
  done
done 2>&1 | tee $O/bf16_grouped_ab.txt
