#!/bin/bash
# GPU-box step: the soaks (the compat ladder against the compiled reference; every algorithm against the ladder) and the
# single-rank RCCL rehearsal line with stdout claimed.
mkdir -p gpurun_out/r03/lines
f=bench_n1_nccl_single_rank_rehearsal
ECSIMD_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03/lines/$f.json 2> gpurun_out/r03/lines/$f.err
echo "rehearsal rc=$? stdout lines: $(wc -l < gpurun_out/r03/lines/$f.json)"
python3 tools/soak.py 22 8 > gpurun_out/r03/soak_vs_reference.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r03/soak_vs_reference.txt
python3 tools/soak_windowed.py 22 64 > gpurun_out/r03/soak_across_algorithms.txt 2>&1; echo "soak_windowed rc=$?"; tail -2 gpurun_out/r03/soak_across_algorithms.txt
