#!/bin/bash
# GPU run 22 of round 6: hardware counters of single launches of the shipped four-wave F(4x4) kernel (48-channel and
# 192-channel classes, B = 64): MFMA busy, wait / issue split, LDS, L1 / L2 -- VERDICT r5 item 1's counter check.
set -u
O=gpurun_out/r06a22
mkdir -p $O
for c in "56,48,48,3" "14,192,192,3"; do
  bash tools/pmc_conv.sh $O/pmc_${c//,/_} --tiles wino4 --iters 6 --filter $c > /dev/null 2>&1
done
rm -rf $O/pmc_*/p*/ 2>/dev/null
for d in $O/pmc_*; do echo "#### $d"; cat $d/summary.txt; done | tee $O/pmc_all.txt | grep -E "####|MfmaUtil|SQ_WAIT|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_ANY|VALUBusy|GRBM|MFMA_BUSY|SQ_WAVES"
