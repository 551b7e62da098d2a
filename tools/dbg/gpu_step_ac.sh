#!/bin/bash
# GPU-box step: soak across algorithms on the final tree (the constant-time comb included), 2 x 24 batches of 2^22.
mkdir -p gpurun_out/r03
timeout -k 10 1100 python3 tools/soak_windowed.py 22 24 > gpurun_out/r03/soak_across_algorithms_final.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r03/soak_across_algorithms_final.txt
