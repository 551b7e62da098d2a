#!/bin/bash
# GPU-box step: final tree (GLV constant-time loop reordered): suite, traffic and line of the secp256k1 constant-time variable base.
mkdir -p gpurun_out/r03/lines
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_final12.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_final12.txt
bash tools/profile_traffic.sh r03k windowed-ct > gpurun_out/r03/traffic_r03k.log 2>&1; tail -1 gpurun_out/r03/traffic_r03k.log
python3 tools/summarize_traffic.py r03 gpurun_out/traffic_r03k
run() { local f="$1"; shift; python3 bench.py "$@" > "gpurun_out/r03/lines/$f.json" 2> "gpurun_out/r03/lines/$f.err"; echo "$f rc=$? $(python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print('%.3f M/s  frac %.3f traffic %.0f MB' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['traffic']/1e6))" 2>/dev/null)"; }
run bench_n1_windowed_constant_time_secp256k1 --steps 10 --warmup 2 --workload windowed-ct --curve secp256k1
run bench_n1_windowed_constant_time --steps 10 --warmup 2 --workload windowed-ct
