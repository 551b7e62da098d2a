#!/bin/bash
# GPU-box step: the constant-time comb as shipped (5-bit windows on both curves): suite, traffic, lines.
mkdir -p gpurun_out/r03/lines
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_final13.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_final13.txt
bash tools/profile_traffic.sh r03l fixed-base-ct > gpurun_out/r03/traffic_r03l.log 2>&1; tail -1 gpurun_out/r03/traffic_r03l.log
python3 tools/summarize_traffic.py r03 gpurun_out/traffic_r03l
run() { local f="$1"; shift; python3 bench.py "$@" > "gpurun_out/r03/lines/$f.json" 2> "gpurun_out/r03/lines/$f.err"; echo "$f rc=$? $(python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print('%.3f M/s  frac %.3f traffic %.0f MB' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['traffic']/1e6))" 2>/dev/null)"; }
run bench_n1_fixed_base_constant_time --steps 20 --warmup 2 --workload fixed-base-ct
run bench_n1_fixed_base_constant_time_secp256k1 --steps 20 --warmup 2 --workload fixed-base-ct --curve secp256k1
python3 tools/soak_windowed.py 22 4 > gpurun_out/r03/soak_ct5.txt 2>&1; tail -1 gpurun_out/r03/soak_ct5.txt
