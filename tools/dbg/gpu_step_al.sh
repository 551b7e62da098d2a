#!/bin/bash
# GPU-box step: secp256k1 constant-time comb: 4-bit (shipped) against 6-bit windows with 768-thread workgroups (148 VGPRs, no spill).
mkdir -p gpurun_out/r03
ECSIMD_HIP_LIBRARY=$PWD/build/variants/ct6k768/libecsimd_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time_fixed or exceptional" > gpurun_out/r03/pytest_ct6k.txt 2>&1; rc=$?; tail -3 gpurun_out/r03/pytest_ct6k.txt
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_variants.py "--workload fixed-base-ct --curve secp256k1 --steps 20 --warmup 2" ct_4bit=base ct_6bit_768=build/variants/ct6k768/libecsimd_hip.so > gpurun_out/r03/ab_ct6k768.txt 2>&1; cat gpurun_out/r03/ab_ct6k768.txt
python3 tools/ab_variants.py "--workload fixed-base-ct --steps 20 --warmup 2" p256_ct_6bit_1024=base p256_ct_6bit_768=build/variants/ct6k768/libecsimd_hip.so >> gpurun_out/r03/ab_ct6k768.txt 2>&1; tail -2 gpurun_out/r03/ab_ct6k768.txt
