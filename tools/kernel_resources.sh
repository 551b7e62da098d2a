#!/bin/bash
# profiles/<round>/kernel_resource_usage.txt: VGPRs, scratch, occupancy and spills of every kernel, from the compiler's own
# remarks (cross-compiles here, no GPU).     tools/kernel_resources.sh r03
set -e
cd "$(dirname "$0")/../ecsimd_amd/csrc"
out=../../profiles/${1:?round}/kernel_resource_usage.txt
: > "$out"
for f in k_*.hip; do
  echo "== $f" >> "$out"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | python3 ../../tools/resusage.py >> "$out"
done
wc -l "$out"
