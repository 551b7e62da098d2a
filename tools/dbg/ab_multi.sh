#!/bin/bash
# several A/B pairs on one box: bash tools/dbg/ab_multi.sh <other library> "<workload curve>" ...
other=$1; shift
out=gpurun_out/ab_multi; mkdir -p $out; : > $out/ab.txt
for spec in "$@"; do
  set -- $spec
  echo "== $1 $2" >> $out/ab.txt
  timeout -k 10 300 python tools/ab_variants.py "--workload $1 --curve $2 --global-log2-batch 22 --steps 5 --warmup 1" new=base other=$other >> $out/ab.txt 2>&1 || exit 1
done
cat $out/ab.txt
