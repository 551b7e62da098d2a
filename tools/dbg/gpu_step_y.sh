#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_large.py -m gpu -q -x -k thread > gpurun_out/r03/pytest_threads.txt 2>&1; rc=$?; tail -25 gpurun_out/r03/pytest_threads.txt; exit $rc
