#!/bin/bash
# round 3 run M: grouped branch levels in the engine (default) vs per-layer launches on four streams
set -u
mkdir -p gpurun_out
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/m_bench_grouped.json 2>/dev/null; cut -c90-260 gpurun_out/m_bench_grouped.json
SHAPY_GROUP_BRANCHES=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | cut -c90-260
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --single-stream | cut -c90-260
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/m_gpu_tests.log 2>&1; tail -5 gpurun_out/m_gpu_tests.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/m_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/m_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r['Kernel_Name'].replace('void shapy::', '').split('(')[0][:70]
    agg[k][0] += 1; agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f'{k:72s} n={v[0]:5d} total {v[1] / 5:9.1f} us/forward  avg {v[1] / v[0]:7.1f} us')
PY
rm -rf gpurun_out/m_trace
