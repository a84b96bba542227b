#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bvh or mesh_to_mesh" 2>&1 | tail -3
timeout 300 python tools/bvh_timing.py 2>&1 | grep -v amdgpu.ids
timeout 300 python bench.py --workload bvh --meshes 1000 --steps 10 --warmup 2 > gpurun_out/p_bench_bvh.json 2>gpurun_out/p_bench_bvh.err; cut -c1-330 gpurun_out/p_bench_bvh.json; tail -2 gpurun_out/p_bench_bvh.err
