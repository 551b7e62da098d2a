#!/bin/bash
# GPU-box step: rocprofv3 kernel trace + PMC passes (tools/profile.sh) of the window workloads of round 3: the constant-time forms and the signed comb.
for W in windowed-ct fixed-base-ct fixed-base-signed; do
  bash tools/profile.sh r03_$W --workload $W --global-log2-batch 22 > gpurun_out/prof_r03_$W.log 2>&1; echo "$W rc=$?"; tail -1 gpurun_out/prof_r03_$W.log
done
