#!/bin/bash
# GPU run K of round 4: the betas all-gather as a direct ncclAllGather on the compute stream (no c10d stream)
set -u
mkdir -p gpurun_out/r04k
O=gpurun_out/r04k
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "forced_gather" 2>&1 | tail -3
for m in rccl work; do
  timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also --force-gather --gather-mode $m 2>$O/err_$m.txt | grep '^{' | tail -1 > $O/bench_force_gather_$m.json
  python -c "import json; d=json.load(open('$O/bench_force_gather_$m.json')); print('force-gather $m', round(d['value'],1), d.get('rccl_ranks'), d['force_gather']['mode'])" || tail -5 $O/err_$m.txt
done
timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 > $O/bench_plain.json
python -c "import json; d=json.load(open('$O/bench_plain.json')); print('plain', round(d['value'],1))"
timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also --force-gather 2>/dev/null | grep '^{' | tail -1 > $O/bench_force_gather_default.json
python -c "import json; d=json.load(open('$O/bench_force_gather_default.json')); print('force-gather default', round(d['value'],1), d['force_gather'])"
