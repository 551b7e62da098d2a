#!/bin/bash
# GPU-box step: the round's committed artefacts -- every bench line, the -m gpu suite, the headline's rocprof passes.
mkdir -p gpurun_out/r03
bash tools/bench_all.sh r03 2>&1 | tee gpurun_out/r03/bench_all.log
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_3.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_3.txt
bash tools/profile.sh r03_ladder > gpurun_out/r03/profile_ladder.log 2>&1; tail -3 gpurun_out/r03/profile_ladder.log
