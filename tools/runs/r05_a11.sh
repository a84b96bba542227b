#!/bin/bash
# GPU run 11 of round 5: upsample-scatter layers with UPS workgroups per tile (one row of every block each): conv
# tests (bit-identical arithmetic), class times of the scatter layers, bf16 bs 32 / f32 bs 1, 8, 64 end to end.
set -u
mkdir -p gpurun_out/r05a11
O=$GRAFT_REPO_ROOT/gpurun_out/r05a11
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "conv_kernel or bf16 or features_vs_reference or full_forward_bs64 or event_driven" 2>&1 | tail -4 | tee $O/tests.txt
timeout 300 python tools/conv_bench.py --tiles auto --iters 20 2>&1 | grep -E " u[248] " | cut -c1-120 | tee $O/scatter_classes_f32_b64.txt
timeout 300 python tools/conv_bench.py --tiles auto --iters 20 --dtype bf16 --batch 32 2>&1 | grep -E " u[248] " | cut -c1-120 | tee $O/scatter_classes_bf16_b32.txt
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", round(d["ms_per_step"],3), "ms per step")'; }
for rep in 1 2; do
  echo "rep $rep bf16 B=32: $(bench --dtype bf16 --batch 32)"
  echo "rep $rep f32 B=1: $(bench --batch 1)"
  echo "rep $rep f32 B=8: $(bench --batch 8)"
  echo "rep $rep f32 B=64: $(bench)"
done | tee $O/bench.txt
