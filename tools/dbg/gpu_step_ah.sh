#!/bin/bash
# GPU-box step: the constant-time window loop reading two whole entries (one 128-byte line) per round: parity, rate at 2 and 3 waves per SIMD, traffic.
mkdir -p gpurun_out/r03/lines
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time" > gpurun_out/r03/pytest_ctv.txt 2>&1; rc=$?; tail -4 gpurun_out/r03/pytest_ctv.txt
[ $rc -eq 0 ] || exit $rc
{
python3 tools/ab_variants.py "--workload windowed-ct --steps 8 --warmup 2" waves2=base waves3=build/variants/ctv3/libecsimd_hip.so
python3 tools/ab_variants.py "--workload windowed-ct --curve secp256k1 --steps 8 --warmup 2" waves2=base waves3=build/variants/ctv3/libecsimd_hip.so
} > gpurun_out/r03/ab_ctv_waves.txt 2>&1; cat gpurun_out/r03/ab_ctv_waves.txt
bash tools/profile_traffic.sh r03h windowed-ct > gpurun_out/r03/traffic_r03h.log 2>&1; tail -2 gpurun_out/r03/traffic_r03h.log
python3 tools/summarize_traffic.py r03 gpurun_out/traffic_r03h
