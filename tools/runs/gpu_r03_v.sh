#!/bin/bash
set -u
timeout 300 python tools/host_enqueue_time.py 2>&1 | grep -v amdgpu.ids | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "event_driven" 2>&1 | grep -v amdgpu.ids | head -60 | cut -c1-200
