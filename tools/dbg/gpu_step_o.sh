#!/bin/bash
# GPU-box step: the signed 7-bit LDS kernel at 256 / 512 / 768 / 1024 threads per workgroup (1 / 2 / 3 / 4 waves per SIMD; one workgroup per CU).
mkdir -p gpurun_out/r03
python3 tools/ab_variants.py "--workload fixed-base-signed --steps 20 --warmup 2" sw768=base sw256=build/variants/sw256/libecsimd_hip.so sw512=build/variants/sw512/libecsimd_hip.so sw1024=build/variants/sw1024/libecsimd_hip.so > gpurun_out/r03/ab_sw_block.txt 2>&1
cat gpurun_out/r03/ab_sw_block.txt
