#!/bin/bash
set -u
for p in 0 1; do echo "lane priorities=$p: $(SHAPY_LANE_PRIO=$p timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"; done
for p in 0 1; do echo "lane priorities=$p: $(SHAPY_LANE_PRIO=$p timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-190)"; done
