#!/bin/bash
set -u
echo "dag on source lanes:"; for d in 1 0 1; do echo "dag=$d: $(SHAPY_DAG=$d timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"; done
echo "dag on aux lanes:"; echo "aux: $(SHAPY_DAG_AUX=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"
echo "aux + GPU_MAX_HW_QUEUES=8: $(GPU_MAX_HW_QUEUES=8 SHAPY_DAG_AUX=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"
echo "source lanes + GPU_MAX_HW_QUEUES=8: $(GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "event_driven" 2>&1 | grep -v amdgpu.ids | tail -3
