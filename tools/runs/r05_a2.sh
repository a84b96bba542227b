#!/bin/bash
# GPU run 2 of round 5: first hardware run of the F(4x4) split-K kernels (csrc/conv_wino4.hip, template parameter S).
# (a) parity: the new kernel tests + the backbone tests under the new default plan (384@7x7: S = 2);
# (b) isolated class times S = 1 / 2 / 3 / 4 (tools/conv_bench.py, tiles wino4kS);
# (c) end to end: SHAPY_W4_KSPLIT policies, interleaved repetitions.
set -u
mkdir -p gpurun_out/r05a2
O=gpurun_out/r05a2
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "split_k or winograd4_kernel or winograd4_features or full_forward_bs64 or event_driven_plan_equals" 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python tools/conv_bench.py --tiles wino4,wino4k2,wino4k3,wino4k4 --iters 20 2>&1 | grep -E "^ *(7|14) .*(384-> 384|192-> 192|512-> 512)" | cut -c1-170 | tee $O/split_classes.txt
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", d["betas_sha1"])'; }
for rep in 1 2; do
  for pol in "" "384@4:2" "384@4:3" "384@4:4" "384@4:2,192@16:2"; do
    echo "rep $rep ksplit='$pol': $(SHAPY_W4_KSPLIT="$pol" bench)"
  done
  echo "rep $rep ksplit='384@4:2,512@4:2' n64=512: $(SHAPY_W4_KSPLIT='384@4:2,512@4:2' SHAPY_WINO4_N64=512 bench)"
  echo "rep $rep ksplit='384@4:2,512@4:4' n64=512: $(SHAPY_W4_KSPLIT='384@4:2,512@4:4' SHAPY_WINO4_N64=512 bench)"
done 2>&1 | tee $O/split_bench.txt
