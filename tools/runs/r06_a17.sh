#!/bin/bash
# GPU run 17 of round 6: per-tap address arithmetic also with three chunks in flight (the bf16 layers with
# Cin % 32 == 0): kernel tests, configs[2]'s shard, bs 64, and the f32 headline unchanged.
set -u
O=gpurun_out/r06a17
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "kernel_bf16 or igemm_split or bf16_features or conv_kernel_vs_float64 or concat_offset_and_inplace or keep_nan" 2>&1 | tail -4 | tee $O/tests.txt
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", round(d["ms_per_step"],3), "ms per step", d["betas_sha1"])'; }
for rep in 1 2; do
  echo "rep $rep bf16 B=32: $(bench --dtype bf16 --batch 32)   one at a time: $(bench --dtype bf16 --batch 32 --pipeline off)"
  echo "rep $rep bf16 B=64: $(bench --dtype bf16 --batch 64)"
  echo "rep $rep f32 B=64: $(bench)"
done 2>&1 | tee $O/ab.txt
