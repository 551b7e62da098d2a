#!/bin/bash
# several A/B pairs on one box: current build against build/ab_prev
out=gpurun_out/ab_multi; mkdir -p $out; : > $out/ab.txt
for spec in "windowed p256" "windowed-ct p256" "windowed secp256k1" "ladder secp256k1" "ladder p256"; do
  set -- $spec
  echo "== $1 $2" >> $out/ab.txt
  timeout -k 10 300 python tools/ab_variants.py "--workload $1 --curve $2 --global-log2-batch 22 --steps 5 --warmup 1" new=base prev=build/ab_prev/libecsimd_hip.so >> $out/ab.txt 2>&1 || exit 1
done
cat $out/ab.txt
