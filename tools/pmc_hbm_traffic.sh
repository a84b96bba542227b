#!/bin/bash
# HBM traffic of the backbone's conv kernels from rocprofv3 PMC counters (separate passes for
# FETCH_SIZE and WRITE_SIZE, counters only).  Writes <out>.json in the layout bench.py reads
# (profiles/*_pmc_hbm_traffic*.json).   usage: tools/pmc_hbm_traffic.sh gpurun_out/pmc_hbm [dtype] [algo] [batch] [plan]
# plan: "single" (default: --single-stream, per-kernel times add up) or "multi" = the plan that is benchmarked
# (four lanes, event-driven; one forward at a time so that the launch count per forward is exact -- the
# pipelined loop runs the same kernels).  Counter collection serialises dispatches either way: byte and
# busy-cycle TOTALS per forward are what these passes give, not concurrency.
set -u
OUT=$1; DT=${2:-f32}; ALGO=${3:-winograd}; BATCH=${4:-64}; PLAN=${5:-single}
PLANFLAGS="--single-stream"; [ "$PLAN" = multi ] && PLANFLAGS="--pipeline off"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
# one launch per layer and no calibration pass: FORWARDS x 333 conv launches, nothing else
export SHAPY_WINO_GUARD=0 SHAPY_GROUP_BRANCHES=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/$tag" -- \
      python "$ROOT/bench.py" --steps 2 --warmup 1 $PLANFLAGS --no-cpu-baseline --no-also --dtype $DT --algo $ALGO --batch $BATCH \
      > "$ROOT/$OUT/$tag.log" 2>&1
done
cd "$ROOT"
python - "$OUT" "$DT" "$ALGO" "$BATCH" "$PLAN" <<'PY'
import csv, glob, json, sys, collections
out, dt, algo, batch, plan = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
FORWARDS = 3          # --steps 2 --warmup 1; every forward = 330 backbone convs + the SMPL-X GEMMs
tot = collections.defaultdict(float)
n = collections.defaultdict(int)
per = collections.defaultdict(dict)          # dispatch order within its pass -> counters
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'conv_' in r['Kernel_Name']]
    order = {}
    for r in sorted(rows, key=lambda r: int(r['Dispatch_Id'])):
        tot[r['Counter_Name']] += float(r['Counter_Value'])
        n[r['Counter_Name']] += 1
        k = order.setdefault(r['Counter_Name'], [0])
        key = k[0] % 333                        # position inside one forward (333 conv launches)
        k[0] += 1
        if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            e = per[key]
            e['kernel'] = r['Kernel_Name'].replace('void shapy::', '').split('(')[0]
            e['grid'] = r.get('Grid_Size', '')
            e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value']) * 1024 / 3
with open(out + '_per_launch.csv', 'w') as fh:
    fh.write('launch,kernel,grid,read_bytes_x2,write_bytes\n')
    for key in sorted(per):
        e = per[key]
        fh.write(f"{key},{e.get('kernel')},{e.get('grid')},{2 * e.get('FETCH_SIZE', 0):.0f},{e.get('WRITE_SIZE', 0):.0f}\n")
fw = {k: float(FORWARDS) for k in n}
res = {'note': 'tools/pmc_hbm_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / '
               'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE in separate passes over python bench.py '
               '--steps 2 --warmup 1 --single-stream, conv kernels only.  FETCH/WRITE_SIZE unit: KB. '
               'gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM '
               'section); the corrected read traffic lies between FETCH and 2*FETCH.',
       'dtype': dt, 'conv_algo': algo, 'launches_seen': dict(n), 'forwards_seen': fw,
       'plan': 'four lanes (event-driven), one forward at a time: bench.py --pipeline off' if plan == 'multi'
               else 'bench.py --single-stream'}
if 'FETCH_SIZE' in tot and 'WRITE_SIZE' in tot:
    rd = tot['FETCH_SIZE'] * 1024 / fw['FETCH_SIZE']
    wr = tot['WRITE_SIZE'] * 1024 / fw['WRITE_SIZE']
    res['hbm_bytes_per_backbone_forward'] = {'read_as_reported': rd, 'write': wr, 'as_reported': rd + wr,
                                             'fetch_x2_corrected': 2 * rd + wr, 'batch': batch, 'size': 224,
                                             'dtype': dt, 'algo': algo}
if 'SQ_VALU_MFMA_BUSY_CYCLES' in tot:
    res['mfma'] = {'SQ_VALU_MFMA_BUSY_CYCLES_sum': tot['SQ_VALU_MFMA_BUSY_CYCLES'],
                   'GRBM_GUI_ACTIVE_sum_over_8_xcd': tot['GRBM_GUI_ACTIVE'],
                   'busy_cycles_per_forward': tot['SQ_VALU_MFMA_BUSY_CYCLES'] / fw['SQ_VALU_MFMA_BUSY_CYCLES']}
json.dump(res, open(out + '.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$ROOT/$OUT"/*/
