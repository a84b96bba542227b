#!/bin/bash
# Run O: all-K Winograd staging (Cin = 48 / 64) and the level-parallel kinematic chain.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== winograd + body tests"
timeout 900 python -m pytest tests -q -m gpu -x -k "winograd or smplx or pose or lut or full_forward or body" 2>&1 | tail -5
echo "== conv bench 48/64 classes: all-K vs chunked"
for f in 56,48,48,3 56,64,64,3; do
  timeout 300 python tools/conv_bench.py --tiles wino,winochunk --filter $f 2>/dev/null | grep -E "^ *[0-9]+ +[0-9]+->" | cut -c1-160
done
echo "== bench default"
timeout 600 python bench.py 2>gpurun_out/o_bench_default.err > gpurun_out/o_bench_default.json; cut -c1-400 gpurun_out/o_bench_default.json
echo "== kernel stats single stream"
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/o_prof -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --single-stream > $R/gpurun_out/o_prof.log 2>&1)
f=$(find gpurun_out/o_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/o_kernel_stats_regressor_singlestream.csv; head -12 "$f" | cut -c1-150
rm -rf gpurun_out/o_prof
