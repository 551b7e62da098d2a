#!/bin/bash
# GPU-box step: ALG_CONSTANT_TIME served by a 6-bit comb (43 windows x 32 entries, all read) instead of the 4-bit one (64 x 8): parity first.
mkdir -p gpurun_out/r03
ECSIMD_HIP_LIBRARY=$PWD/build/variants/ct6/libecsimd_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time or exceptional" > gpurun_out/r03/pytest_ct6.txt 2>&1; rc=$?; tail -5 gpurun_out/r03/pytest_ct6.txt
[ $rc -eq 0 ] || exit $rc
{
python3 tools/ab_variants.py "--workload fixed-base-ct --steps 20 --warmup 2" ct_4bit=base ct_6bit=build/variants/ct6/libecsimd_hip.so
python3 tools/ab_variants.py "--workload fixed-base-ct --curve secp256k1 --steps 20 --warmup 2" ct_4bit=base ct_6bit=build/variants/ct6/libecsimd_hip.so
} > gpurun_out/r03/ab_ct6.txt 2>&1
cat gpurun_out/r03/ab_ct6.txt
