#!/bin/bash
# round 3 run B: where does the multi-stream wall time go (per-phase timeline), F(4x4) on the 7x7 maps
# end to end, per-kernel MFMA busy of the three F(4x4) classes
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/b_trace_ms -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/b_trace_ms.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/b_trace_ss -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --single-stream > $R/gpurun_out/b_trace_ss.log 2>&1
cd $R
echo "== timeline multi-stream"
timeout 200 python tools/timeline.py gpurun_out/b_trace_ms > gpurun_out/b_timeline_ms.txt 2>&1; tail -12 gpurun_out/b_timeline_ms.txt
echo "== timeline single-stream"
timeout 200 python tools/timeline.py gpurun_out/b_trace_ss > gpurun_out/b_timeline_ss.txt 2>&1; tail -12 gpurun_out/b_timeline_ss.txt
timeout 200 python tools/timeline.py gpurun_out/b_trace_ms --verbose > gpurun_out/b_timeline_ms_verbose.txt 2>&1
echo "== wino4 from 7 px end to end"
timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --wino4-min-hw 7 | cut -c90-200
timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline | cut -c90-200
echo "== per-kernel MFMA busy, F(4x4) classes stand-alone"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/b_pmc_w4 -- python $R/tools/conv_bench.py --tiles wino4 --iters 4 > $R/gpurun_out/b_pmc_w4.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/b_pmc_w4/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void shapy::', '')[:40] + ' grid=' + r.get('Grid_Size', '')
        if 'wino4' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, {c: f'{v:.4g}' for c, v in m.items()},
          'mfma busy / (gui_active x 128 simd x n_xcd?)', m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(m.get('GRBM_GUI_ACTIVE', 1), 1) / 128)
PY
rm -rf gpurun_out/b_trace_ms gpurun_out/b_trace_ss gpurun_out/b_pmc_w4
