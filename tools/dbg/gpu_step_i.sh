#!/bin/bash
# GPU-box step: full suite, the signed fixed-base line, secondary kernels, traffic of the signed workload -- after its odd-digit rewrite.
mkdir -p gpurun_out/r03/lines
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_final4.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_final4.txt
python3 bench.py --steps 20 --warmup 2 --workload fixed-base-signed > gpurun_out/r03/lines/bench_n1_fixed_base_signed7.json 2> gpurun_out/r03/lines/bench_n1_fixed_base_signed7.err; echo "line rc=$?"
python3 tools/bench_kernels.py > gpurun_out/r03/secondary_kernels.json 2> gpurun_out/r03/secondary_kernels.txt; echo "secondary rc=$?"
bash tools/profile_traffic.sh r03c fixed-base-signed > gpurun_out/r03/traffic_r03c.log 2>&1; tail -2 gpurun_out/r03/traffic_r03c.log
python3 tools/soak_windowed.py 22 16 > gpurun_out/r03/soak_across_algorithms_2.txt 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r03/soak_across_algorithms_2.txt
