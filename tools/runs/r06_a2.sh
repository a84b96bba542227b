#!/bin/bash
# GPU run 2 of round 6: where do the chunks of the four-wave F(4x4) kernel spend their time?
# (a) timing builds (SHAPY_W4Q_DBG masks: 1 transform VALU, 2 patch loads, 4 V writes, 8 filter refills, 16 V reads);
# (b) counters of the 192-channel and 48-channel launches, new kernel next to the 3 + 1-wave kernel.
set -u
O=gpurun_out/r06a2
mkdir -p $O
R=$PWD
cls() { grep -E "^ *(56   48->  48|28   96->  96|14  192-> 192|  7  384-> 384).*r1" | cut -c1-150; }
echo "== product" | tee $O/ablations.txt
timeout 200 python tools/conv_bench.py --tiles wino4,wino4old,wino4k2 --iters 20 2>&1 | cls | tee -a $O/ablations.txt
for d in 1 2 4 7 8 15 31; do
  echo "== dbg=$d" | tee -a $O/ablations.txt
  SHAPY_HIP_LIB=$R/tools/bin/libshapy_dbg$d.so timeout 200 python tools/conv_bench.py --tiles wino4,wino4k2 --iters 20 2>&1 | cls | tee -a $O/ablations.txt
done
for t in wino4 wino4old; do
  for c in "14,192,192,3" "56,48,48,3"; do
    bash tools/pmc_conv.sh $O/pmc_${t}_${c//,/_} --tiles $t --iters 6 --filter $c > /dev/null 2>&1
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $R/$O/icache -- python $R/tools/conv_bench.py --tiles wino4,wino4old --iters 6 --filter 14,192,192,3 > $R/$O/icache.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r06a2/icache/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/r06a2/icache_summary.txt', 'w') as fo:
    for k, d in agg.items():
        fo.write(k + '\n')
        for c, v in sorted(d.items()):
            fo.write(f'   {c:30s} mean {sum(v)/len(v):.4g} n={len(v)}\n')
print(open('gpurun_out/r06a2/icache_summary.txt').read())
PY
rm -rf $O/icache $O/pmc_*/p*/ 2>/dev/null
for d in $O/pmc_*; do echo "#### $d"; cat $d/summary.txt; done > $O/pmc_all.txt
tail -5 $O/icache.log
