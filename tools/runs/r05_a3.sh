#!/bin/bash
# GPU run 3 of round 5: (a) the split-K kernel tests with their diagnostics (the first case failed in run 2 and
# only the tail of the log came back), the SMPL-X odd-batch test, the four gather modes on one rank;
# (b) backbone latency at B = 1 / 8 with and without the split; (c) a verbose timeline of the default plan
# (384@7x7: S = 2) -> module tails; (d) the betas gather on the executor's lane-1 stream vs on the compute
# stream vs none (world-size-1 communicator, --force-gather).
set -u
mkdir -p gpurun_out/r05a3
O=$GRAFT_REPO_ROOT/gpurun_out/r05a3
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s --tb=short -k "split_k" 2>&1 | grep -v "^$" | tail -60 > $O/tests_split.txt
tail -25 $O/tests_split.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -k "smplx_forward_odd or rccl_forced or winograd4_kernel" 2>&1 | tail -15 | tee $O/tests_other.txt
bench() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", d["betas_sha1"])'; }
for rep in 1 2; do for b in 1 8; do for pol in "" "384@4:2" "384@4:4"; do
  echo "rep $rep B=$b ksplit='$pol': $(SHAPY_W4_KSPLIT="$pol" bench --batch $b)"
done; done; done 2>&1 | tee $O/split_small_batch.txt
for rep in 1 2; do
  echo "rep $rep plain: $(bench)"
  echo "rep $rep force-gather lane: $(bench --force-gather --gather-mode lane)"
  echo "rep $rep force-gather rccl: $(bench --force-gather --gather-mode rccl)"
done 2>&1 | tee $O/gather_ab.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-also > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python tools/timeline.py $O/prof --verbose > $O/timeline_verbose.txt 2>$O/timeline_err.txt; tail -3 $O/timeline_err.txt
python tools/module_tails.py $O/timeline_verbose.txt | tee $O/module_tails.txt
rm -rf $O/prof
