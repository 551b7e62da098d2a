#!/bin/bash
# GPU-box step: counters of the two LDS comb kernels side by side (why does the signed 7-bit kernel take 3.70 cycles per VALU instruction
# and the 4-bit one 3.36 with the same addition code?).
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p $REPO/gpurun_out/r03/comb_pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $REPO/gpurun_out/r03/comb_pmc/list_avail.txt 2>&1
for WL in fixed-base fixed-base-signed; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
    i=$((i+1))
    rocprofv3 --pmc $C --output-format csv -d "$REPO/gpurun_out/r03/comb_pmc/${WL}_$i" -- python3 "$REPO/bench.py" --workload $WL --global-log2-batch 22 --steps 2 --warmup 0 --no-cpu-baseline > "$REPO/gpurun_out/r03/comb_pmc/${WL}_$i.log" 2>&1 || tail -3 "$REPO/gpurun_out/r03/comb_pmc/${WL}_$i.log"
    echo "$WL pass $i done"
  done
done
