#!/bin/bash
# round 5, GPU run 28: the full bench line of the opt-in mode (--head-gemm bf16x6) on the final tree
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/a28
timeout 90 python bench.py --head-gemm bf16x6 --no-also --no-cpu-baseline > gpurun_out/a28/bench_head_gemm_bf16x6.json 2> gpurun_out/a28/err.txt
python -c "
import json; d=json.loads(open('gpurun_out/a28/bench_head_gemm_bf16x6.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['config']['head_gemm_arithmetic'], d['betas_sha1'])"
