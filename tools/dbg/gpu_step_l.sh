#!/bin/bash
# GPU-box step: the signed 7-bit LDS kernel with one persistent workgroup per CU (table loaded once) against one workgroup per 768 lanes.
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "windowed or comb or fixed or config3 or base" > gpurun_out/r03/pytest_swin.txt 2>&1; rc=$?; tail -3 gpurun_out/r03/pytest_swin.txt
[ $rc -eq 0 ] || exit $rc
{
python3 tools/ab_variants.py "--workload fixed-base-signed --steps 20 --warmup 2" persistent=base per_768_lanes=build/variants/swin_np/libecsimd_hip.so
python3 tools/ab_variants.py "--workload fixed-base-signed --curve secp256k1 --steps 20 --warmup 2" persistent=base per_768_lanes=build/variants/swin_np/libecsimd_hip.so
} > gpurun_out/r03/ab_swin_persist.txt 2>&1
cat gpurun_out/r03/ab_swin_persist.txt
