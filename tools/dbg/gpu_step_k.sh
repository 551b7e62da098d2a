#!/bin/bash
# GPU-box step: workgroup size / occupancy of the 4-bit LDS kernel after its odd-digit rewrite (secp256k1 allocates 146 VGPRs: one 512-thread
# workgroup per CU), and the headline at the per-GPU sizes of an 8-, 4- and 2-GPU strong-scaling run.
mkdir -p gpurun_out/r03
{
python3 tools/ab_variants.py "--workload fixed-base --steps 20 --warmup 2" wb512=base wb256=build/variants/wb256/libecsimd_hip.so wb768=build/variants/wb768/libecsimd_hip.so
python3 tools/ab_variants.py "--workload fixed-base --curve secp256k1 --steps 20 --warmup 2" wb512=base wb256=build/variants/wb256/libecsimd_hip.so wb768=build/variants/wb768/libecsimd_hip.so occ4=build/variants/occ4/libecsimd_hip.so
} > gpurun_out/r03/ab_wblock.txt 2>&1
cat gpurun_out/r03/ab_wblock.txt
for b in 21 22 23; do python3 bench.py --no-cpu-baseline --global-log2-batch $b --steps 20 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('2^$b: %.3f M/s  %.3f ms/step  kernel %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']))"; done > gpurun_out/r03/shard_sizes.txt 2>&1
cat gpurun_out/r03/shard_sizes.txt
