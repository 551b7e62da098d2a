#!/bin/bash
# GPU-box step: secp256k1 constant-time comb with 5-bit windows (52 x 16 entries, 53 KB, three 256-thread workgroups per CU) against the 4-bit one.
mkdir -p gpurun_out/r03
ECSIMD_HIP_LIBRARY=$PWD/build/variants/ct5/libecsimd_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time_fixed or exceptional" > gpurun_out/r03/pytest_ct5.txt 2>&1; rc=$?; tail -3 gpurun_out/r03/pytest_ct5.txt
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_variants.py "--workload fixed-base-ct --curve secp256k1 --steps 20 --warmup 2" ct_4bit=base ct_5bit_256=build/variants/ct5/libecsimd_hip.so > gpurun_out/r03/ab_ct5.txt 2>&1; cat gpurun_out/r03/ab_ct5.txt
