#!/bin/bash
# GPU run S of round 4: 64-channel N tile of the F(4x4) kernel (four multiplying waves): kernel tests, class
# timings against F(2x2), end to end with layer1 / head 3x3 layers on it
set -u
mkdir -p gpurun_out/r04s
O=gpurun_out/r04s
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_kernel or winograd4_concat" 2>&1 | tail -4
timeout 300 python tools/conv_bench.py --tiles wino,wino4 --filter 56,64,64,3 --iters 20 2>&1 | grep "wino"
timeout 300 python tools/conv_bench.py --tiles wino,wino4 --filter 7,512,512,3 --iters 20 2>&1 | grep "wino"
for v in 0 1; do
  echo "SHAPY_WINO4_N64=$v: $(SHAPY_WINO4_N64=$v timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone, frac", round(d["roofline"]["frac"],4), "betas", d.get("parity",{}).get("betas_l2"))')"
done | tee $O/n64_ab.txt
SHAPY_WINO4_N64=1 timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "full_forward_bs64 or winograd4_features or features_256 or guard" 2>&1 | tail -3
