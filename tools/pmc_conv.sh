#!/bin/bash
# Hardware-counter passes over one conv class of tools/conv_bench.py (one rocprofv3 run per group;
# counters only, no tracing).  usage: tools/pmc_conv.sh <outdir> <conv_bench args...>
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "VmemLatency LdsLatency MemUnitStalled" \
           "LdsUtil LdsBankConflict MfmaUtil VALUBusy" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/p$i" -- \
      python "$ROOT/tools/conv_bench.py" "$@" > "$ROOT/$OUT/p$i.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'conv_' not in k:
            continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for k, d in agg.items():
        fo.write(k + '\n')
        for c, v in sorted(d.items()):
            fo.write(f'   {c:40s} mean {sum(v)/len(v):.4g}  n={len(v)}\n')
print(open(out + '/summary.txt').read())
PY
