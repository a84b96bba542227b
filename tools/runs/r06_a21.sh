#!/bin/bash
# GPU run 21 of round 6: the head's 512 -> 512 3x3 layers on the F(4x4) kernel with a partly filled last N tile
# (ten full 48-channel tiles + one of 32), split over two workgroups: kernel tests, class time, end to end against F(2x2).
set -u
O=gpurun_out/r06a21
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_kernel or winograd4_split_k or winograd4_concat or keep_nan" 2>&1 | tail -5 | tee $O/tests_kernel.txt
timeout 300 python tools/conv_bench.py --tiles wino,wino4,wino4k2 --iters 20 --filter 7,512,512,3 2>&1 | grep "512->" | cut -c1-170 | tee $O/class_512.txt
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms", d["betas_sha1"], round(d["roofline"]["frac"],4))'; }
for rep in 1 2 3; do
  echo "rep $rep F(4x4) 512: $(bench)   one at a time: $(bench --pipeline off)"
  echo "rep $rep F(2x2) 512: $(SHAPY_WINO4_PARTIAL= bench)   one at a time: $(SHAPY_WINO4_PARTIAL= bench --pipeline off)"
done 2>&1 | tee $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_features or full_forward_bs64 or features_256 or full_forward_vs_reference or guard" 2>&1 | tail -4 | tee $O/tests_backbone.txt
