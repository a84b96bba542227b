#!/bin/bash
# GPU run 25 of round 6: VERDICT r5's side experiment as a K-loop skeleton (tools/w4_bf16x3_skeleton.hip): the F(4x4)
# position-GEMMs on the bf16 pipe with the exact 3-way split against the f32 MFMAs -- filter ring from L2, V from LDS, MFMAs only.
set -u
mkdir -p gpurun_out/r06a25
timeout 120 tools/bin/w4_bf16x3_skeleton 2>&1 | tee gpurun_out/r06a25/w4_bf16x3_skeleton.txt
