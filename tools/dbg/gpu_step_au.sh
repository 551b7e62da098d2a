#!/bin/bash
# GPU-box step: constant-time combs as shipped (P-256 6-bit, secp256k1 5-bit): parity; P-256 5-bit at 256 threads against the shipped 6-bit at 1024.
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time_fixed or exceptional" > gpurun_out/r03/pytest_ct_ship.txt 2>&1; rc=$?; tail -3 gpurun_out/r03/pytest_ct_ship.txt
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_variants.py "--workload fixed-base-ct --steps 20 --warmup 2" p256_6bit_1024=base p256_5bit_256=build/variants/ctp5/libecsimd_hip.so > gpurun_out/r03/ab_ctp5.txt 2>&1; cat gpurun_out/r03/ab_ctp5.txt
