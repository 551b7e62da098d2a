#!/bin/bash
# GPU-box step: stand-alone ZDAU kernel capped at 128 registers (4 waves per SIMD; 2^20 points = exactly 4 rounds of waves instead of 5.33).
mkdir -p gpurun_out/r03
{ echo "# 3 waves per SIMD (143 / 135 VGPRs, no spill)"; python3 tools/dbg/zdau_ab.py; echo "# 4 waves per SIMD (128 VGPRs, 16 / 8 spills)"; ECSIMD_HIP_LIBRARY=$PWD/build/variants/zdau4/libecsimd_hip.so python3 tools/dbg/zdau_ab.py; } > gpurun_out/r03/ab_zdau_kernel_waves.txt 2>&1
cat gpurun_out/r03/ab_zdau_kernel_waves.txt
