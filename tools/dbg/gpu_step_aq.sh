#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time" > gpurun_out/r03/pytest_ct_final.txt 2>&1; rc=$?; tail -4 gpurun_out/r03/pytest_ct_final.txt; exit $rc
