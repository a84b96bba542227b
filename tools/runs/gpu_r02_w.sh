#!/bin/bash
# round 2, GPU run W (the last ~9 GPU-minutes of the round): first run of the Winograd F(4x4,3x3)
# kernel (csrc/conv_wino4.hip).  Order = value per second: parity tests of the new kernel, end-to-end
# A/B against the F(2x2) default, then the whole suite + smoke under the best configuration as the
# default (SHAPY_CONV_ALGO / SHAPY_WINO4_MIN_HW), then per-class timings, profile, ring variants.
# Every step writes its own file under gpurun_out/ so that a cut-off run keeps what it finished.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
S0=$SECONDS
stamp() { echo "[t=$((SECONDS - S0))s] $*"; }

stamp "canary: one tiny launch of the new kernel (a barrier mismatch would hang, not fail)"
timeout 240 python -m pytest -q -x --tb=short \
    "tests/test_gpu_parity.py::test_conv_winograd4_kernel_vs_float64[(1, 8, 8, 16, 48, False, False)]" \
    > gpurun_out/w_canary.log 2>&1
crc=$?
tail -n 3 gpurun_out/w_canary.log
if [ $crc -eq 124 ] || [ $crc -eq 137 ]; then
  echo "rc=124 canary timed out: F(4x4) kernel skipped" > gpurun_out/w_wino4_tests.log
  stamp "CANARY HUNG -- the rest of the run uses the F(2x2) default only"
  WINO4_DEAD=1
else
  WINO4_DEAD=0
  stamp "wino4 tests"
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "winograd4" --tb=short -s \
      > gpurun_out/w_wino4_tests.log 2>&1
  echo "rc=$?" >> gpurun_out/w_wino4_tests.log
fi
grep -E "passed|failed|error|F\(4x4\) layers|bs64 f32\+winograd4 (features|betas|vertices)" gpurun_out/w_wino4_tests.log | tail -n 12

stamp "bench A/B"
for cfg in "winograd 28" "winograd4 28" "winograd4 14"; do
  set -- $cfg
  [ "$1" = winograd4 ] && [ $WINO4_DEAD -eq 1 ] && continue
  SHAPY_CONV_ALGO=$1 SHAPY_WINO4_MIN_HW=$2 timeout 200 python bench.py --steps 15 --warmup 4 \
      --no-cpu-baseline 2> gpurun_out/w_bench_$1_$2.err > gpurun_out/w_bench_$1_$2.json
  python - "$1" "$2" <<'EOF'
import json, sys
try:
    d = json.load(open(f'gpurun_out/w_bench_{sys.argv[1]}_{sys.argv[2]}.json'))
    print(f"  {sys.argv[1]:>10s} min_hw {sys.argv[2]}: {d['value']:.0f} images/s, backbone "
          f"{d['roofline']['ms_per_launch_group']:.2f} ms, algo {d['config']['conv_algo']}")
except Exception as e:
    print('  bench failed:', sys.argv[1:], e)
EOF
done

# best configuration whose parity tests passed -> default for the rest of the run
eval $(python - <<'EOF'
import json, re
best, arg = 0.0, ('winograd', 28)
ok4 = False
try:
    log = open('gpurun_out/w_wino4_tests.log').read()
    ok4 = 'rc=0' in log and ' failed' not in log
except OSError:
    pass
for algo, hw in (('winograd', 28), ('winograd4', 28), ('winograd4', 14)):
    if algo == 'winograd4' and not ok4:
        continue
    try:
        v = json.load(open(f'gpurun_out/w_bench_{algo}_{hw}.json'))['value']
    except Exception:
        continue
    if v > best * (1.0 if algo == 'winograd' else 1.02):     # F(4x4) has to win by > 2 %
        best, arg = v, (algo, hw)
print(f'export SHAPY_CONV_ALGO={arg[0]} SHAPY_WINO4_MIN_HW={arg[1]}')
EOF
)
stamp "chosen default: $SHAPY_CONV_ALGO (F(4x4) from $SHAPY_WINO4_MIN_HW px)"
echo "$SHAPY_CONV_ALGO $SHAPY_WINO4_MIN_HW" > gpurun_out/w_chosen.txt

stamp "full suite under that default"
timeout 600 python -m pytest tests -q -m gpu --tb=short > gpurun_out/w_all_tests.log 2>&1
tail -n 4 gpurun_out/w_all_tests.log
stamp "per-class timings"
[ $WINO4_DEAD -eq 0 ] && timeout 200 python tools/conv_bench.py --tiles wino,wino4 --wino4-min-hw 7 --iters 10 \
    > gpurun_out/w_conv_bench_wino_vs_wino4.txt 2>&1
grep -E "wino4" gpurun_out/w_conv_bench_wino_vs_wino4.txt | cut -c1-150
stamp "driver bench"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/w_bench_default.err \
    > gpurun_out/w_bench_default.json
cut -c1-420 gpurun_out/w_bench_default.json

for v in 6 9; do
  [ $WINO4_DEAD -eq 1 ] && continue
  SHAPY_HIP_LIB=$R/shapy_amd/csrc/libshapy_hip_w4r$v.so timeout 120 python tools/conv_bench.py \
      --tiles wino4 --wino4-min-hw 14 --iters 10 > gpurun_out/w_conv_bench_wino4_ring$v.txt 2>&1
  echo "ring $v:"; grep -E "wino4" gpurun_out/w_conv_bench_wino4_ring$v.txt | cut -c1-110
done

stamp "smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
stamp "rocprof kernel stats"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/w_prof \
    -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --single-stream \
    > $R/gpurun_out/w_prof.log 2>&1)
f=$(find gpurun_out/w_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/w_kernel_stats_singlestream.csv && head -8 "$f" | cut -c1-150
rm -rf gpurun_out/w_prof
stamp "hipGraph replay at bs 64"
timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --graph on 2>/dev/null \
    > gpurun_out/w_bench_graph_on.json; cut -c90-260 gpurun_out/w_bench_graph_on.json
stamp "done"
