#!/bin/bash
# GPU-box step: GLV constant-time loop with each table line requested right before its doubling (no read pending during the additions: 29 spills) against the cross-iteration prefetch (53).
mkdir -p gpurun_out/r03
ECSIMD_HIP_LIBRARY=$PWD/build/variants/glvnew/libecsimd_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time_variable" > gpurun_out/r03/pytest_glvnew.txt 2>&1; rc=$?; tail -3 gpurun_out/r03/pytest_glvnew.txt
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_variants.py "--workload windowed-ct --curve secp256k1 --steps 8 --warmup 2" prefetch_across_windows_53_spills=base line_per_doubling_29_spills=build/variants/glvnew/libecsimd_hip.so > gpurun_out/r03/ab_glv_ct_order.txt 2>&1; cat gpurun_out/r03/ab_glv_ct_order.txt
