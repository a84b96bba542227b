#!/bin/bash
# GPU run P of round 4: the SMPL-X layer as one C call (shapy_smplx_forward_f32): parity tests, the layer's lines
set -u
mkdir -p gpurun_out/r04p
O=gpurun_out/r04p
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "smplx or full_forward or shipped or demo or lut or forced_gather" 2>&1 | tail -4
timeout 600 python -m pytest tests -m gpu -x -q -k "not gpu_parity" 2>&1 | tail -3
for b in 4 64; do
  timeout 200 python bench.py --workload smplx --batch $b --steps 50 --warmup 10 2>/dev/null | grep '^{' | tail -1 > $O/bench_smplx_b$b.json
  python -c "import json; d=json.load(open('$O/bench_smplx_b$b.json')); print('smplx b$b', round(d['value'],1), 'bodies/s', round(d['ms_per_step']*1e3,1), 'us/call', round(d['roofline']['frac'],4))"
done
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_default_nocpu.json
python -c "import json; d=json.load(open('$O/bench_default_nocpu.json')); print(round(d['value'],1), round(d['ms_per_step'],3), {k:(round(v.get('value',0),1), round(v.get('roofline',{}).get('frac',0),4)) for k,v in d['also'].items()})"
