#!/bin/bash
# GPU run X of round 4: the stem kernel with 4 pixels x 8 channels per thread (packed FMAs): parity, its time
set -u
mkdir -p gpurun_out/r04x
O=$PWD/gpurun_out/r04x
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "features or full_forward_vs_reference or bf16" 2>&1 | tail -3
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o stem -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-also > $O/bench_prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
grep -E "stem_conv|Name" "$f" | cut -c1-200 | tee $O/stem_kernel_stats.txt
rm -rf $O/prof
echo "bench: $(timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "img/s", round(d["roofline"]["ms_per_launch_group"],3), "ms backbone", d.get("parity"))')" | tee $O/bench.txt
