#!/bin/bash
# GPU-box step: signed 7-bit LDS kernel, spill-free candidates against the spilling 1024-thread defaults.
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "windowed or comb or fixed or config3 or base" > gpurun_out/r03/pytest_swin.txt 2>&1; rc=$?; tail -3 gpurun_out/r03/pytest_swin.txt
[ $rc -eq 0 ] || exit $rc
{
echo "# P-256: default = 1024 threads + prefetch (19 spills); no_prefetch_1024 = first window peeled, 119 VGPRs, 0 spills"
python3 tools/ab_variants.py "--workload fixed-base-signed --steps 20 --warmup 2" default=base no_prefetch_1024=build/variants/nopf/libecsimd_hip.so no_prefetch_768=build/variants/nopf768/libecsimd_hip.so prefetch_768=build/variants/pf768/libecsimd_hip.so
echo "# secp256k1: default = 1024 threads, no prefetch, first window peeled (18 spills); 768-thread forms have no spills"
python3 tools/ab_variants.py "--workload fixed-base-signed --curve secp256k1 --steps 20 --warmup 2" default=base no_prefetch_768=build/variants/nopf768/libecsimd_hip.so prefetch_768=build/variants/pf768/libecsimd_hip.so
} > gpurun_out/r03/ab_swin_spill_free.txt 2>&1
cat gpurun_out/r03/ab_swin_spill_free.txt
