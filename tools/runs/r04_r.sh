#!/bin/bash
# GPU run R of round 4: SMPL-X argument glue as one launch (shapy_smplx_prepare_f32) + three K chunks in flight
# for the pose blend GEMM: parity tests of everything that goes through SMPLX.forward, then the layer's lines
set -u
mkdir -p gpurun_out/r04r
O=gpurun_out/r04r
timeout 900 python -m pytest tests -m gpu -x -q -k "smplx or full_forward or shipped or demo or lut or metrics or attributes or evaluate or virtual or smoke" 2>&1 | tail -4
for pd in 0 1; do for b in 4 64; do
  SHAPY_SMPLX_PD1=$( [ $pd = 1 ] && echo 1 || echo "" ) timeout 200 env $( [ $pd = 1 ] && echo SHAPY_SMPLX_PD1=1 || echo SHAPY_DUMMY=1 ) python bench.py --workload smplx --batch $b --steps 100 --warmup 20 2>/dev/null | grep '^{' | tail -1 > $O/bench_smplx_b${b}_pd1_$pd.json
  python -c "import json; d=json.load(open('$O/bench_smplx_b${b}_pd1_$pd.json')); print('smplx b$b pd1=$pd', round(d['value'],1), 'bodies/s', round(d['ms_per_step']*1e3,1), 'us/call', round(d['roofline']['frac'],4))"
done; done
timeout 400 python bench.py --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['value'],1), round(d['ms_per_step'],3))"
