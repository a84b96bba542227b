#!/bin/bash
# GPU run 27 of round 5 (closing build): the whole GPU suite + smoke on the final tree
set -u
mkdir -p gpurun_out/r05a27
O=$GRAFT_REPO_ROOT/gpurun_out/r05a27
timeout 330 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/gpu_tests_tail.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 | tee $O/smoke.txt
