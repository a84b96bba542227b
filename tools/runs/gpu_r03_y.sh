#!/bin/bash
set -u
mkdir -p gpurun_out
for v in 1 0 1 0; do echo "dag_balance=$v: $(SHAPY_DAG_BALANCE=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"; done
timeout 900 bash tools/pmc_hbm_traffic.sh gpurun_out/y_pmc_hbm_traffic_winograd4 f32 winograd4 | grep -A8 hbm_bytes
echo "bf16 b64: $(timeout 200 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"
echo "bf16 b32: $(timeout 200 python bench.py --dtype bf16 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"
