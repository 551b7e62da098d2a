#!/bin/bash
# GPU-box step: final LDS kernels -- traffic of the signed workload, full suite, then (they read the committed traffic of the PREVIOUS pass
# for the 4-bit lines, unchanged kernels) the two signed lines; the signed lines' traffic field is patched from this pass afterwards.
mkdir -p gpurun_out/r03/lines
bash tools/profile_traffic.sh r03e fixed-base-signed > gpurun_out/r03/traffic_r03e.log 2>&1; tail -2 gpurun_out/r03/traffic_r03e.log
python3 tools/summarize_traffic.py r03 gpurun_out/traffic_r03e
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_final6.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_final6.txt
run() { local f="$1"; shift; python3 bench.py "$@" > "gpurun_out/r03/lines/$f.json" 2> "gpurun_out/r03/lines/$f.err"; echo "$f rc=$? $(python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print('%.3f M/s  frac %.3f traffic %.0f MB' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['traffic']/1e6))" 2>/dev/null)"; }
run bench_n1_fixed_base_signed7 --steps 20 --warmup 2 --workload fixed-base-signed
run bench_n1_fixed_base_signed7_secp256k1 --steps 20 --warmup 2 --workload fixed-base-signed --curve secp256k1
python3 tools/bench_kernels.py > gpurun_out/r03/secondary_kernels.json 2> gpurun_out/r03/secondary_kernels.txt; echo "secondary rc=$?"
cp profiles/pmc_traffic.json gpurun_out/r03/pmc_traffic_after_r03e.json
