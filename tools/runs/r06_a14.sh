#!/bin/bash
# GPU run 14 of round 6: step latency by batch (one forward at a time, as profiles/r05m_bench_by_batch.txt) on the
# four-wave kernel + per-tap implicit GEMM, and the split-K policies of the small-batch buckets re-checked.
set -u
O=gpurun_out/r06a14
mkdir -p $O
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also --pipeline off "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", round(d["ms_per_step"],3), "ms per step")'; }
for b in 1 4 8 16 32 64; do echo "f32 B=$b: $(bench --batch $b)"; done | tee $O/bench_by_batch.txt
echo "bf16 B=32: $(bench --dtype bf16 --batch 32)" | tee -a $O/bench_by_batch.txt
for pol in "" "384@4:2" "384@4:4,192@16:2" "384@4:4,192@16:4,96@49:2"; do
  for b in 1 8 32; do echo "ksplit='$pol' B=$b: $(SHAPY_W4_KSPLIT="$pol" bench --batch $b)"; done
done | tee $O/ksplit_small_batch.txt
