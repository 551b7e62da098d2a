#!/bin/bash
# GPU-box step: signed 7-bit LDS kernel, 1024-thread workgroups (4 waves per SIMD, 128 registers, spills) with and without the LDS prefetch.
mkdir -p gpurun_out/r03
{
python3 tools/ab_variants.py "--workload fixed-base-signed --steps 20 --warmup 2" sw768=base sw1024=build/variants/sw1024/libecsimd_hip.so sw1024_no_prefetch=build/variants/sw1024nopf/libecsimd_hip.so
python3 tools/ab_variants.py "--workload fixed-base-signed --curve secp256k1 --steps 20 --warmup 2" sw768=base sw1024=build/variants/sw1024/libecsimd_hip.so sw1024_no_prefetch=build/variants/sw1024nopf/libecsimd_hip.so
} > gpurun_out/r03/ab_sw_block2.txt 2>&1
cat gpurun_out/r03/ab_sw_block2.txt
