#!/bin/bash
# GPU run 12 of round 5: first hardware run of split-K on the implicit-GEMM kernel: kernel tests, class times, policy
# sweeps (bf16 bs 32 / 64; f32 small batches).
set -u
mkdir -p gpurun_out/r05a12
O=$GRAFT_REPO_ROOT/gpurun_out/r05a12
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -k "igemm_split_k or winograd_guard or calibrat" 2>&1 | grep -v "^$" | tail -14 | tee $O/tests.txt
true
timeout 600 python tools/direct_ksplit_sweep.py --dtype bf16 --batches 32,64 2>&1 | grep -v amdgpu.ids | tee $O/sweep_bf16.txt
timeout 600 python tools/direct_ksplit_sweep.py --dtype f32 --batches 1,8,32,64 2>&1 | grep -v amdgpu.ids | tee $O/sweep_f32.txt
