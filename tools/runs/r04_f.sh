#!/bin/bash
# GPU run F of round 4: start stagger of a CU's second workgroup in the F(4x4) kernels (per layer and
# persistent / grouped): sweep, then end to end.
set -u
mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
echo "== conv_bench: stagger sweep (units of 128 clocks)"
timeout 400 python tools/conv_bench.py --tiles wino4,wino4s8,wino4s16,wino4s24,wino4s32,wino4s48,wino4s64 --iters 20 2>&1 | grep -E "wino4" | tee $O/conv_bench_stagger.txt
echo "== grouped / persistent: stagger sweep"
for s in 0 16 32 48; do echo "-- stagger $s"; timeout 300 python tools/wino4g_check.py --bench --stagger $s 2>&1 | grep -v amdgpu.ids; done | tee $O/w4g_stagger.txt
echo "== end to end"
for s in 0 16 32 48; do
  fl=$(python -c "print(hex($s << 24))")
  for f in "" "--group-branches on"; do
    echo "bench stagger $s $f: $(timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-also --tile-flags $fl $f 2>&1 | grep -v amdgpu.ids | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone, frac", round(d["roofline"]["frac"],4))')"
  done
done | tee $O/bench_stagger.txt
