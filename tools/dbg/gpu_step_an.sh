#!/bin/bash
# GPU-box step: the whole suite once more (x-only test with 4 106 oracle lanes per form; launcher edits).
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_final10.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_final10.txt
