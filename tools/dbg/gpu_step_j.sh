#!/bin/bash
# GPU-box step: the two-rank rehearsal of bench.py on one GPU (gloo), as a test and as a recorded line at 2^24.
mkdir -p gpurun_out/r03/lines
timeout -k 10 900 python -m pytest tests/test_bench_gpu.py -m gpu -q -x > gpurun_out/r03/pytest_bench_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/r03/pytest_bench_gpu.txt
[ $rc -eq 0 ] || exit $rc
ECSIMD_BENCH_REHEARSE_ONE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03/lines/bench_rehearsal_2_ranks_one_gpu.json 2> gpurun_out/r03/lines/bench_rehearsal_2_ranks_one_gpu.err; echo "rehearsal rc=$?"
tail -c 600 gpurun_out/r03/lines/bench_rehearsal_2_ranks_one_gpu.err
