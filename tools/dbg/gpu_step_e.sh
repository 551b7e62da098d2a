#!/bin/bash
# GPU-box step: A/B of the conditional-swap form; the single-rank rehearsal line with the gathered-shard sample check.
mkdir -p gpurun_out/r03/lines
python3 tools/ab_variants.py "--steps 8 --warmup 2" xor_and=base cndmask=build/variants/cswap_cndmask/libecsimd_hip.so xor_and_again=base cndmask_again=build/variants/cswap_cndmask/libecsimd_hip.so | tee gpurun_out/r03/ab_cswap_cndmask.txt
f=bench_n1_nccl_single_rank_rehearsal
ECSIMD_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03/lines/$f.json 2> gpurun_out/r03/lines/$f.err
echo "rehearsal rc=$? stdout lines: $(wc -l < gpurun_out/r03/lines/$f.json)"; python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/$f.json')); print(d['value'], d['config']['gather'].get('sample_check'))"
