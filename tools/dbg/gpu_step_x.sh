#!/bin/bash
# GPU-box step: the beyond-4-GiB test on its own.
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_large.py -m gpu -q -x > gpurun_out/r03/pytest_large.txt 2>&1; rc=$?; tail -25 gpurun_out/r03/pytest_large.txt; exit $rc
