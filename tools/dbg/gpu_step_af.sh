#!/bin/bash
# GPU-box step: ALG_CONSTANT_TIME on a variable base (every entry of the lane's window table read in every window): parity, then rate.
mkdir -p gpurun_out/r03/lines
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time" > gpurun_out/r03/pytest_ctv.txt 2>&1; rc=$?; tail -12 gpurun_out/r03/pytest_ctv.txt
[ $rc -eq 0 ] || exit $rc
run() { local f="$1"; shift; python3 bench.py "$@" > "gpurun_out/r03/lines/$f.json" 2> "gpurun_out/r03/lines/$f.err"; echo "$f rc=$? $(python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print('%.3f M/s  frac %.3f' % (d['value']/1e6, d['roofline']['frac']))" 2>/dev/null)"; tail -2 gpurun_out/r03/lines/$f.err | cut -c1-300; }
run bench_n1_windowed_constant_time --steps 8 --warmup 2 --workload windowed-ct --no-cpu-baseline
run bench_n1_windowed_constant_time_secp256k1 --steps 8 --warmup 2 --workload windowed-ct --curve secp256k1 --no-cpu-baseline
