#!/bin/bash
# GPU run 9 of round 5: the whole GPU suite with the batch-bucketed split-K policy (every network test at B <= 8 now
# runs the 7x7 / 14x14 / 28x28 branches split), smoke, backbone latency through bench.py at B = 1 / 8 / 32.
set -u
mkdir -p gpurun_out/r05a9
O=$GRAFT_REPO_ROOT/gpurun_out/r05a9
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/gpu_tests_tail.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -12 | tee $O/smoke.txt
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", round(d["ms_per_step"],3), "ms per step", d["config"]["wino4_ksplit"])'; }
for b in 1 8 32 64; do echo "B=$b: $(bench --batch $b)"; done | tee $O/bench_by_batch.txt
