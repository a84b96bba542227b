#!/bin/bash
# GPU run Q of round 4: the event-driven plan as an explicitly built hipGraph: parity, then timing against the
# eager forward and the captured barrier plan at B = 1 / 8 / 64 (f32) and at configs[2]'s shard (bf16, bs 32)
set -u
mkdir -p gpurun_out/r04q
O=gpurun_out/r04q
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "explicit_graph or graph_replay" 2>&1 | tail -15 | tee $O/tests.txt
for b in 1 8 64; do for g in off explicit on; do
  echo "f32 B=$b graph=$g: $(timeout 300 python bench.py --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-also --graph $g 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone")' 2>&1 | tail -1)"
done; done | tee $O/graph_ab.txt
for g in off explicit; do
  echo "bf16 B=32 graph=$g: $(timeout 300 python bench.py --dtype bf16 --batch 32 --steps 30 --warmup 8 --no-cpu-baseline --no-also --graph $g 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone")' 2>&1 | tail -1)"
done | tee -a $O/graph_ab.txt
