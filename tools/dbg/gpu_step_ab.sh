#!/bin/bash
# GPU-box step: the whole suite, smoke and the secondary kernels on the final tree.
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_final7.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_final7.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 tools/bench_kernels.py > gpurun_out/r03/secondary_kernels.json 2> gpurun_out/r03/secondary_kernels.txt; echo "secondary rc=$?"
