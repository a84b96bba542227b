#!/bin/bash
# Run P: producer / consumer Winograd kernel -- correctness, then per-class timing.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== ps kernel tests"
timeout 300 python -m pytest tests -q -m gpu -x -k "producer_consumer" 2>&1 | tail -5
echo "== conv bench"
for f in 56,48,48,3 28,96,96,3 14,192,192,3 7,384,384,3; do
  timeout 200 python tools/conv_bench.py --tiles wino,winops,winops3 --filter $f 2>/dev/null | grep -E "^ *[0-9]+ +[0-9]+->" | cut -c1-170
done
