#!/bin/bash
# GPU-box step: after the odd-digit 4-bit comb and the k* substitution -- full suite, the fixed-base bench lines, their traffic passes.
mkdir -p gpurun_out/r03/lines
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_final3.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_final3.txt
run() { local f="$1"; shift; python3 bench.py "$@" > "gpurun_out/r03/lines/$f.json" 2> "gpurun_out/r03/lines/$f.err"; echo "$f rc=$? $(python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print('%.3f M/s  frac %.3f' % (d['value']/1e6, d['roofline']['frac']))" 2>/dev/null)"; }
run bench_n1_fixed_base --steps 20 --warmup 2 --workload fixed-base
run bench_n1_fixed_base_secp256k1 --steps 20 --warmup 2 --workload fixed-base --curve secp256k1
run bench_n1_fixed_base_big20 --steps 20 --warmup 2 --workload fixed-base-big
run bench_n1_fixed_base_signed7 --steps 20 --warmup 2 --workload fixed-base-signed
bash tools/profile_traffic.sh r03b fixed-base fixed-base-big > gpurun_out/r03/traffic_r03b.log 2>&1; tail -4 gpurun_out/r03/traffic_r03b.log
python3 tools/ab_variants.py "--workload fixed-base --steps 20 --warmup 2" odd_with_kstar=base unsigned16=build/variants/fixed4_old/libecsimd_hip.so | tee gpurun_out/r03/ab_fixed4_odd.txt
python3 tools/ab_variants.py "--workload fixed-base --curve secp256k1 --steps 20 --warmup 2" odd_with_kstar=base unsigned16=build/variants/fixed4_old/libecsimd_hip.so | tee -a gpurun_out/r03/ab_fixed4_odd.txt
