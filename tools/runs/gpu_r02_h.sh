#!/bin/bash
# round 2, GPU run H: Winograd phase timing, SMPL-X (LBS) kernel stats
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== winograd phase timing"; timeout 900 python tools/wino_timing.py 2>&1 | tail -8
echo "== rocprof smplx B=64"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/h_prof_smplx -- python $R/bench.py --workload smplx --batch 64 --steps 20 --warmup 5 > $R/gpurun_out/h_prof_smplx.log 2>&1)
f=$(find gpurun_out/h_prof_smplx -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/h_kernel_stats_smplx_b64.csv; head -12 "$f" | cut -c1-90,140-220
f=$(find gpurun_out/h_prof_smplx -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'conv_igemm' in n:
        n='conv_igemm GEMM grid '+r.get('Grid_Size', r.get('Grid_Size_X','?'))
    agg[n[:70]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    v=v[len(v)//3:]
    print(f'{len(v):4d} calls  avg {sum(v)/len(v):8.1f} us  {k}')
PY
rm -rf gpurun_out/h_prof_smplx
echo "== bench smplx"; for b in 4 64; do timeout 300 python bench.py --workload smplx --batch $b 2>/dev/null > gpurun_out/h_bench_smplx_b$b.json; cut -c1-300 gpurun_out/h_bench_smplx_b$b.json; python -c "import json; d=json.load(open('gpurun_out/h_bench_smplx_b$b.json')); print(d['roofline'])"; done
