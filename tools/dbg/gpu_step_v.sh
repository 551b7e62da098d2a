#!/bin/bash
# GPU-box step: a longer soak of the final tree against the compiled reference (both ladders), 2 x 16 batches of 2^22.
mkdir -p gpurun_out/r03
timeout -k 10 1150 python3 tools/soak.py 22 16 > gpurun_out/r03/soak_vs_reference_long.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r03/soak_vs_reference_long.txt
