#!/bin/bash
# GPU-box step: the four fixed-base bench lines on one box with the final LDS kernels (they read the refreshed profiles/pmc_traffic.json).
mkdir -p gpurun_out/r03/lines
run() { local f="$1"; shift; python3 bench.py "$@" > "gpurun_out/r03/lines/$f.json" 2> "gpurun_out/r03/lines/$f.err"; echo "$f rc=$? $(python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print('%.3f M/s  frac %.3f' % (d['value']/1e6, d['roofline']['frac']))" 2>/dev/null)"; }
run bench_n1_fixed_base --steps 20 --warmup 2 --workload fixed-base
run bench_n1_fixed_base_secp256k1 --steps 20 --warmup 2 --workload fixed-base --curve secp256k1
run bench_n1_fixed_base_big20 --steps 20 --warmup 2 --workload fixed-base-big
run bench_n1_fixed_base_signed7 --steps 20 --warmup 2 --workload fixed-base-signed
run bench_n1_fixed_base_signed7_secp256k1 --steps 20 --warmup 2 --workload fixed-base-signed --curve secp256k1
