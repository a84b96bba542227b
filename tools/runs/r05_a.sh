#!/bin/bash
# Prepared at the end of round 4 for the FIRST GPU run of round 5: the fuse_add plans (HighResolutionNet.fuse_add =
# 1 | 2; DESIGN section 8, profiles/r04o_module_tails.txt).  Written and CPU-verified (tests/test_plan_replay.py)
# after round 4's GPU budget was spent; the kernel (csrc/hrnet_ops.hip: fuse_add_kernel) has never run.
# Expected: stage-3 modules 1,000 -> ~850 us, stage-4 modules 1,450 -> 1,000-1,250 us (0.6-2 ms of the 12.4 ms step).
set -u
mkdir -p gpurun_out/r05a
O=$GRAFT_REPO_ROOT/gpurun_out/r05a
timeout 900 python -m pytest tests/test_zz_fuse_add_gpu.py -q -rA 2>&1 | tail -40 | tee $O/fuse_add_tests.txt
bench1() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", "betas", (d.get("parity") or {}).get("betas_l2"))'; }
for rep in 1 2; do
  echo "rep $rep fuse_add 0: $(bench1 --fuse-add 0)"
  echo "rep $rep fuse_add 1: $(bench1 --fuse-add 1)"
  echo "rep $rep fuse_add 2 dest,dest,mixed: $(bench1 --fuse-add 2 --fuse-chain-lanes dest,dest,mixed)"
  echo "rep $rep fuse_add 2 dest,dest,dest: $(bench1 --fuse-add 2 --fuse-chain-lanes dest,dest,dest)"
  echo "rep $rep fuse_add 2 source,source,source: $(bench1 --fuse-add 2 --fuse-chain-lanes source,source,source)"
  echo "rep $rep fuse_add 1 + launch groups: $(bench1 --fuse-add 1 --group-branches on)"
  echo "rep $rep fuse_add 2 + launch groups: $(bench1 --fuse-add 2 --group-branches on)"
done 2>&1 | tee $O/fuse_add_bench.txt
# where the time goes in each form: kernel trace -> per-op timeline -> per-module lane starts / ends / tails
cd /tmp; export TMPDIR=/tmp
for v in "0" "1" "2 --fuse-chain-lanes dest,dest,mixed"; do
  tag=$(echo $v | cut -c1)
  rm -rf $O/prof$tag
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-also --fuse-add $v > /dev/null 2>&1
  ( cd $GRAFT_REPO_ROOT; timeout 200 python tools/timeline.py $O/prof$tag --verbose --fuse-add $v > $O/timeline_fuse_add_$tag.txt 2>$O/timeline_$tag.err
    python tools/module_tails.py $O/timeline_fuse_add_$tag.txt | tee $O/module_tails_fuse_add_$tag.txt )
  rm -rf $O/prof$tag
done
