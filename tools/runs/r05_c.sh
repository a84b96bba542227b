#!/bin/bash
# Prepared at the end of round 4: CU partition between the lanes (SHAPY_LANE_CU_EIGHTHS="a,b,c,d": eighths of the CUs
# for the streams of lanes 0..3, csrc/hrnet_ops.hip: lane_cu_share).  The hypothesis it tests: the first launch of a
# small-map lane waits 160-300 us for workgroup slots that the large lanes' launches keep refilling
# (profiles/r04o_module_tails.txt); with slots of its own the 7x7 lane of a stage-4 module starts with its module.
set -u
mkdir -p gpurun_out/r05c
O=$GRAFT_REPO_ROOT/gpurun_out/r05c
bench1() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", "betas", (d.get("parity") or {}).get("betas_l2"))'; }
# correctness first: the partition only changes streams, results must be bit-identical
SHAPY_LANE_CU_EIGHTHS=5,1,1,1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "event_driven or features_256 or full_forward_bs64" 2>&1 | tail -2 | tee $O/partition_tests.txt
for rep in 1 2; do
  echo "rep $rep no partition: $(bench1)"
  for sh in 5,1,1,1 4,2,1,1 4,1,1,2 0,0,1,1 0,0,0,1 0,0,0,2; do
    echo "rep $rep SHAPY_LANE_CU_EIGHTHS=$sh: $(SHAPY_LANE_CU_EIGHTHS=$sh bench1)"
  done
done 2>&1 | tee $O/partition_bench.txt
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof
SHAPY_LANE_CU_EIGHTHS=5,1,1,1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-also > /dev/null 2>&1
( cd $GRAFT_REPO_ROOT; timeout 200 python tools/timeline.py $O/prof --verbose > $O/timeline_partition_5111.txt 2>$O/timeline.err
  python tools/module_tails.py $O/timeline_partition_5111.txt | tee $O/module_tails_partition_5111.txt )
rm -rf $O/prof
