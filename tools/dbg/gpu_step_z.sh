#!/bin/bash
# GPU-box step: the constant-time fixed-base kernel -- parity, then its rate beside the default kernel's.
mkdir -p gpurun_out/r03/lines
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time or exceptional" > gpurun_out/r03/pytest_ct.txt 2>&1; rc=$?; tail -15 gpurun_out/r03/pytest_ct.txt
[ $rc -eq 0 ] || exit $rc
run() { local f="$1"; shift; python3 bench.py "$@" > "gpurun_out/r03/lines/$f.json" 2> "gpurun_out/r03/lines/$f.err"; echo "$f rc=$? $(python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print('%.3f M/s  frac %.3f' % (d['value']/1e6, d['roofline']['frac']))" 2>/dev/null)"; }
run bench_n1_fixed_base_constant_time --steps 20 --warmup 2 --workload fixed-base-ct
run bench_n1_fixed_base_constant_time_secp256k1 --steps 20 --warmup 2 --workload fixed-base-ct --curve secp256k1
run bench_n1_fixed_base --steps 20 --warmup 2 --workload fixed-base
run bench_n1_fixed_base_secp256k1 --steps 20 --warmup 2 --workload fixed-base --curve secp256k1
