#!/bin/bash
# GPU run H of round 4: hardware counters of the F(4x4) kernel on the 48-channel 56x56 class (B = 64)
set -u
bash tools/pmc_conv.sh gpurun_out/r04h_pmc48 --tiles wino4 --filter 56,48,48,3 --iters 20 2>&1 | tail -70
