#!/bin/bash
# GPU-box step: the constant-time comb as shipped (6-bit on P-256, 4-bit on secp256k1): parity, traffic pass, lines.
mkdir -p gpurun_out/r03/lines
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_cpp_host_api.py -m gpu -q -x -k "constant_time or exceptional or host_api or cpp" > gpurun_out/r03/pytest_ct.txt 2>&1; rc=$?; tail -5 gpurun_out/r03/pytest_ct.txt
[ $rc -eq 0 ] || exit $rc
bash tools/profile_traffic.sh r03f fixed-base-ct > gpurun_out/r03/traffic_r03f.log 2>&1; tail -2 gpurun_out/r03/traffic_r03f.log
python3 tools/summarize_traffic.py r03 gpurun_out/traffic_r03f
run() { local f="$1"; shift; python3 bench.py "$@" > "gpurun_out/r03/lines/$f.json" 2> "gpurun_out/r03/lines/$f.err"; echo "$f rc=$? $(python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print('%.3f M/s  frac %.3f traffic %.0f MB' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['traffic']/1e6))" 2>/dev/null)"; }
run bench_n1_fixed_base_constant_time --steps 20 --warmup 2 --workload fixed-base-ct
run bench_n1_fixed_base_constant_time_secp256k1 --steps 20 --warmup 2 --workload fixed-base-ct --curve secp256k1
