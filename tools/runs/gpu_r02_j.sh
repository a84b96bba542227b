#!/bin/bash
# round 2, GPU run J: workgroup stagger A/B on the Winograd classes; SMPL-X GEMM durations in order
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for st in 0 2 4 -2; do
  echo "== conv bench, SHAPY_WINO_STAGGER=$st"
  SHAPY_WINO_STAGGER=$st timeout 600 python tools/conv_bench.py --tiles wino --iters 12 2>&1 | grep -E "wino" | cut -c1-100
done
for st in 0 4 -2; do
  echo "== bench default, SHAPY_WINO_STAGGER=$st"
  SHAPY_WINO_STAGGER=$st timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
echo "== smplx kernel trace"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/j_prof_smplx -- python $R/bench.py --workload smplx --batch 64 --steps 10 --warmup 3 > $R/gpurun_out/j_prof_smplx.log 2>&1)
f=$(find gpurun_out/j_prof_smplx -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'shapy' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-14:]:
    print(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  grid {r.get('Grid_Size','?'):>8}  {r['Kernel_Name'][:60]}")
PY
rm -rf gpurun_out/j_prof_smplx
