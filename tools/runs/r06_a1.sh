#!/bin/bash
# GPU run 1 of round 6: first hardware run of the four-multiplying-wave F(4x4) kernel (csrc/conv_wino4q.hip).
# (a) parity: kernel tests vs float64 (plain + split-K), NaN through ReLU, concat offsets;
# (b) isolated class times, new kernel (wino4) against the 3 + 1-wave kernel (wino4old), split variants;
# (c) backbone parity tests + the default bench line.
set -u
O=gpurun_out/r06a1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_kernel or winograd4_split_k or winograd4_concat or keep_nan" 2>&1 | tail -30 | tee $O/tests_kernel.txt
timeout 300 python tools/conv_bench.py --tiles wino4,wino4old,wino4k2,wino4ko2 --iters 20 2>&1 | grep -E "wino4" | cut -c1-200 | tee $O/classes.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd4_features or full_forward_bs64 or event_driven_plan_equals or hrnet_features_256" 2>&1 | tail -8 | tee $O/tests_backbone.txt
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | tee $O/bench.json
