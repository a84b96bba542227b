#!/bin/bash
# GPU run 5 of round 6: the four-wave kernel with the packed (row-pair) input transform: parity, class times and
# end-to-end rate, staging micro-steps spread (group 1) or in blocks of 12 / 36 / 108.
set -u
O=gpurun_out/r06a5
mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_kernel or winograd4_split_k or winograd4_concat or keep_nan" 2>&1 | tail -5 | tee $O/tests_kernel.txt
cls() { grep -E "^ *(56   48->  48|28   96->  96|14  192-> 192|  7  384-> 384|56  256->  48)" | cut -c1-150; }
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms", d["betas_sha1"])'; }
for g in 1 12 36 108; do
  L=$R/tools/bin/libshapy_grp$g.so; [ $g = 1 ] && L=$R/shapy_amd/csrc/libshapy_hip.so
  echo "== group=$g" | tee -a $O/groups.txt
  SHAPY_HIP_LIB=$L timeout 200 python tools/conv_bench.py --tiles wino4,wino4k2 --iters 20 2>&1 | cls | tee -a $O/groups.txt
  echo "bench: $(SHAPY_HIP_LIB=$L bench)   unpipelined: $(SHAPY_HIP_LIB=$L bench --pipeline off)" | tee -a $O/groups.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_features or full_forward_bs64 or event_driven_plan_equals" 2>&1 | tail -4 | tee $O/tests_backbone.txt
