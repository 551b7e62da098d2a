#!/bin/bash
# (record: the fixed-base-signed-ct workload and the ECS_SWIN_BARRIER switch this step used were removed after the measurement -- profiles/r03/ab_constant_time_signed7.txt)
# GPU-box step: ALG_CONSTANT_TIME on the signed 7-bit kernel -- parity, rate; and the secp256k1 signed kernel with a scheduling barrier (10 spills instead of 18).
mkdir -p gpurun_out/r03/lines
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time or exceptional" > gpurun_out/r03/pytest_ct.txt 2>&1; rc=$?; tail -15 gpurun_out/r03/pytest_ct.txt
[ $rc -eq 0 ] || exit $rc
run() { local f="$1"; shift; python3 bench.py "$@" > "gpurun_out/r03/lines/$f.json" 2> "gpurun_out/r03/lines/$f.err"; echo "$f rc=$? $(python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print('%.3f M/s  frac %.3f' % (d['value']/1e6, d['roofline']['frac']))" 2>/dev/null)"; }
run bench_n1_fixed_base_signed7_constant_time --steps 20 --warmup 2 --workload fixed-base-signed-ct
run bench_n1_fixed_base_signed7_constant_time_secp256k1 --steps 20 --warmup 2 --workload fixed-base-signed-ct --curve secp256k1
python3 tools/ab_variants.py "--workload fixed-base-signed --curve secp256k1 --steps 20 --warmup 2" spills18=base sched_barrier_spills10=build/variants/swbar/libecsimd_hip.so > gpurun_out/r03/ab_swin_barrier.txt 2>&1; cat gpurun_out/r03/ab_swin_barrier.txt
