#!/bin/bash
# GPU-box step: LDS comb kernels reading the NEXT window's entry before the current addition (signed 7-bit: default now, two windows per trip;
# 4-bit: -DECS_W4_PREFETCH=1) against reading it where it is used.
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "windowed or comb or fixed or config3 or base" > gpurun_out/r03/pytest_swin.txt 2>&1; rc=$?; tail -3 gpurun_out/r03/pytest_swin.txt
[ $rc -eq 0 ] || exit $rc
{
python3 tools/ab_variants.py "--workload fixed-base-signed --steps 20 --warmup 2" prefetch=base at_use=build/variants/swin_nopf/libecsimd_hip.so
python3 tools/ab_variants.py "--workload fixed-base-signed --curve secp256k1 --steps 20 --warmup 2" prefetch=base at_use=build/variants/swin_nopf/libecsimd_hip.so
python3 tools/ab_variants.py "--workload fixed-base --steps 20 --warmup 2" at_use=base prefetch=build/variants/w4pf/libecsimd_hip.so
python3 tools/ab_variants.py "--workload fixed-base --curve secp256k1 --steps 20 --warmup 2" at_use=base prefetch=build/variants/w4pf/libecsimd_hip.so
} > gpurun_out/r03/ab_lds_prefetch.txt 2>&1
cat gpurun_out/r03/ab_lds_prefetch.txt
