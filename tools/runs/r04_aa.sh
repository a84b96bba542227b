#!/bin/bash
# GPU run AA of round 4 (last minutes of the budget): different issue priorities for the multiply phases of the two
# workgroups of a CU (-DSHAPY_W4_PRIO_PARITY), class times and one bench each
set -u
mkdir -p gpurun_out/r04aa
O=gpurun_out/r04aa
V=$PWD/shapy_amd/csrc/variants/libpp.so
( echo "== variant"; SHAPY_HIP_LIB=$V timeout 100 python tools/conv_bench.py --tiles wino4 --iters 20 2>&1 | grep -E "wino4" | grep "r1\|256->" | cut -c1-100
  echo "== product"; timeout 100 python tools/conv_bench.py --tiles wino4 --iters 20 2>&1 | grep -E "wino4" | grep "r1\|256->" | cut -c1-100 ) | tee $O/prio_parity_classes.txt
for v in $V ""; do
  if [ -n "$v" ]; then export SHAPY_HIP_LIB=$v; else unset SHAPY_HIP_LIB; fi
  echo "bench ${v:-product}: $(timeout 100 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone")')"
done 2>&1 | tee $O/prio_parity_bench.txt
