#!/bin/bash
# GPU-box step: secp256k1 ALG_CONSTANT_TIME on a variable base through the GLV split with complete formulas (k_varwin_mult_glv_ct): parity, rate.
mkdir -p gpurun_out/r03
for V in glvct2 glvct3; do
  ECSIMD_HIP_LIBRARY=$PWD/build/variants/$V/libecsimd_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "constant_time_variable" > gpurun_out/r03/pytest_$V.txt 2>&1; rc=$?; echo "$V parity rc=$rc"; tail -4 gpurun_out/r03/pytest_$V.txt
  [ $rc -eq 0 ] || exit $rc
done
python3 tools/ab_variants.py "--workload windowed-ct --curve secp256k1 --steps 8 --warmup 2" plain_odd_loop=base glv_complete_2waves=build/variants/glvct2/libecsimd_hip.so glv_complete_3waves=build/variants/glvct3/libecsimd_hip.so > gpurun_out/r03/ab_glv_ct.txt 2>&1; cat gpurun_out/r03/ab_glv_ct.txt
