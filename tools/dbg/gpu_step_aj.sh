#!/bin/bash
# GPU-box step: soak across algorithms with both constant-time forms in, 2 x 16 batches of 2^22.
mkdir -p gpurun_out/r03
timeout -k 10 1100 python3 tools/soak_windowed.py 22 16 > gpurun_out/r03/soak_across_algorithms_final2.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r03/soak_across_algorithms_final2.txt
