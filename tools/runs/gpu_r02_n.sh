#!/bin/bash
# round 2, GPU run N: head-layer tile sweep, bf16 status (bench + kernel stats + PMC traffic)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== head layers tile sweep"; timeout 600 python tools/conv_bench.py --tiles auto,32x64,64x64,64x128,128x64,128x128,64x64+bk32 > gpurun_out/conv_bench_r02n_head.txt 2>&1; grep -E "^#|^  7 (1536|2048| 512->2048)" gpurun_out/conv_bench_r02n_head.txt | cut -c1-200
echo "== bf16 bs64"; timeout 300 python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null > gpurun_out/n_bench_bf16_b64.json; cut -c1-300 gpurun_out/n_bench_bf16_b64.json
echo "== bf16 bs32 (configs[2] shard)"; timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline 2>/dev/null > gpurun_out/n_bench_bf16_b32.json; cut -c1-300 gpurun_out/n_bench_bf16_b32.json
echo "== bf16 bs32 graph"; timeout 300 python bench.py --dtype bf16 --batch 32 --graph on --no-cpu-baseline 2>/dev/null > gpurun_out/n_bench_bf16_b32_graph.json; cut -c1-300 gpurun_out/n_bench_bf16_b32_graph.json
echo "== f32 bs32 / bs128 / bs256"; for b in 32 128 256; do timeout 300 python bench.py --batch $b --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($b, round(d['value']), 'img/s', round(d['roofline']['achieved'],1), 'TF')"; done
echo "== rocprof bf16 bs32"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/n_prof_bf16 -- python $R/bench.py --dtype bf16 --batch 32 --steps 5 --warmup 3 --no-cpu-baseline --single-stream > $R/gpurun_out/n_prof_bf16.log 2>&1)
f=$(find gpurun_out/n_prof_bf16 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/n_kernel_stats_bf16_b32_singlestream.csv; head -8 "$f" | cut -c1-120
rm -rf gpurun_out/n_prof_bf16
