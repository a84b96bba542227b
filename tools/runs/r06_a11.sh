#!/bin/bash
# GPU run 11 of round 6: per-tap address arithmetic in the implicit-GEMM loop (all float32 layers), against the
# previous build (1x1 layers only): kernel tests, class times of the direct classes, end to end.
set -u
O=gpurun_out/r06a11
mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_kernel_vs_float64 or igemm_split or concat_offset_and_inplace or keep_nan or hrnet_features_vs_reference_golden or smplx_ops or full_forward_vs_reference_golden" 2>&1 | tail -4 | tee $O/tests.txt
for L in $R/shapy_amd/csrc/libshapy_hip.so $R/tools/bin/libshapy_p11.so; do
  echo "== $(basename $L)" | tee -a $O/classes.txt
  SHAPY_HIP_LIB=$L timeout 300 python tools/conv_bench.py --tiles auto --iters 20 2>&1 | grep -vE "k3 s1 u1" | cut -c1-130 | tee -a $O/classes.txt
done
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms", d["betas_sha1"])'; }
for rep in 1 2; do
  echo "rep $rep per-tap: $(bench)   unpipelined: $(bench --pipeline off)"
  echo "rep $rep previous: $(SHAPY_HIP_LIB=$R/tools/bin/libshapy_p11.so bench)   unpipelined: $(SHAPY_HIP_LIB=$R/tools/bin/libshapy_p11.so bench --pipeline off)"
done 2>&1 | tee $O/ab.txt
