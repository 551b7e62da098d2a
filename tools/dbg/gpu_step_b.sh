#!/bin/bash
# GPU-box step: calibration (with the size-specific request counters) + A/Bs of the GLV window kernel and of ZDAU's Z placement.
bash tools/profile_calib.sh r03b > gpurun_out/calib_r03b.log 2>&1; tail -3 gpurun_out/calib_r03b.log
mkdir -p gpurun_out/r03
python3 tools/ab_variants.py "--workload windowed --curve secp256k1 --steps 5 --warmup 1" base=base glv_late=build/variants/glv_late/libecsimd_hip.so glv_late_2w=build/variants/glv_late_2w/libecsimd_hip.so glv_2w=build/variants/glv_2w/libecsimd_hip.so | tee gpurun_out/r03/ab_glv_window_kernel.txt
python3 tools/ab_variants.py "--steps 5 --warmup 1" z_early=base z_late=build/variants/zlate/libecsimd_hip.so | tee gpurun_out/r03/ab_zdau_z_placement_p256.txt
python3 tools/ab_variants.py "--steps 5 --warmup 1 --curve secp256k1" z_early=base z_late=build/variants/zlate/libecsimd_hip.so | tee gpurun_out/r03/ab_zdau_z_placement_secp256k1.txt
