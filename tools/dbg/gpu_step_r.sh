#!/bin/bash
# GPU-box step: after the LDS kernels' workgroup changes -- traffic passes first (the lines read profiles/pmc_traffic.json, so they need
# a second call), then the full suite, smoke, secondary kernels.
mkdir -p gpurun_out/r03/lines
bash tools/profile_traffic.sh r03d fixed-base fixed-base-signed > gpurun_out/r03/traffic_r03d.log 2>&1; tail -4 gpurun_out/r03/traffic_r03d.log
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_final5.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_final5.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 tools/bench_kernels.py > gpurun_out/r03/secondary_kernels.json 2> gpurun_out/r03/secondary_kernels.txt; echo "secondary rc=$?"
