#!/bin/bash
# round 3 run E: first contact of the persistent grouped F(4x4) kernel (canary under a timeout: a
# barrier-protocol bug hangs instead of failing), then its timing against per-layer launches
set -u
mkdir -p gpurun_out
timeout 150 python tools/wino4g_check.py --canary > gpurun_out/e_canary.txt 2>&1; echo "canary rc=$?"; tail -25 gpurun_out/e_canary.txt
if grep -q "CANARY OK" gpurun_out/e_canary.txt; then
  timeout 300 python tools/wino4g_check.py --bench > gpurun_out/e_bench.txt 2>&1; echo "bench rc=$?"; grep -v amdgpu.ids gpurun_out/e_bench.txt
fi
