#!/bin/bash
# GPU-box step: the table-gather kernels (20-bit fixed base, per-element window tables) capped at 128 registers = 4 waves per SIMD, against 3.
mkdir -p gpurun_out/r03
{
python3 tools/ab_variants.py "--workload fixed-base-big --steps 20 --warmup 2" waves3=base waves4=build/variants/w4all/libecsimd_hip.so
python3 tools/ab_variants.py "--workload fixed-base-big --curve secp256k1 --steps 20 --warmup 2" waves3=base waves4=build/variants/w4all/libecsimd_hip.so
python3 tools/ab_variants.py "--workload windowed --steps 8 --warmup 2" waves3=base waves4=build/variants/w4all/libecsimd_hip.so
python3 tools/ab_variants.py "--workload windowed --curve secp256k1 --steps 8 --warmup 2" waves3=base waves4=build/variants/w4all/libecsimd_hip.so
} > gpurun_out/r03/ab_gather_waves.txt 2>&1
cat gpurun_out/r03/ab_gather_waves.txt
