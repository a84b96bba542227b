#!/bin/bash
# GPU run J of round 4: tile sweep of the direct (implicit GEMM) kernel over every HRNet conv class at B = 64
# + force-gather rehearsal (both issue modes) + HBM traffic PMC pass of the current build
set -u
mkdir -p gpurun_out/r04j
O=gpurun_out/r04j
timeout 600 python tools/conv_bench.py --tiles auto,64x48,64x64,64x96,64x128,128x48,128x64,128x96,128x128,256x48,256x64,32x64 --iters 10 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_tile_sweep.txt | head -60
for m in work side; do
  timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also --force-gather --gather-mode $m 2>/dev/null | grep '^{' | tail -1 > $O/bench_force_gather_$m.json
  python -c "import json; d=json.load(open('$O/bench_force_gather_$m.json')); print('force-gather $m', round(d['value'],1), d.get('rccl_ranks'), d['force_gather'])"
done
timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 > $O/bench_plain.json
python -c "import json; d=json.load(open('$O/bench_plain.json')); print('plain', round(d['value'],1))"
