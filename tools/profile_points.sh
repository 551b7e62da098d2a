#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for tools/point_kernels.py.
# Usage: tools/profile_points.sh <tag> [log2n]        outputs under gpurun_out/prof_<tag>_point/
set -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r02}"
LOG2N="${2:-20}"
OUT="$REPO/gpurun_out/prof_${TAG}_point"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
DRV="$REPO/tools/point_kernels.py"
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$DRV" "$LOG2N" 5 > "$OUT/stats.log" 2>&1 || { tail -20 "$OUT/stats.log"; exit 1; }
grep "us " "$OUT/stats.log"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  NAME=$(echo "$C" | tr ' ' '_' | cut -c1-40)
  echo "== pmc $C"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -- python3 "$DRV" "$LOG2N" 2 > "$OUT/pmc_$NAME.log" 2>&1 || { tail -5 "$OUT/pmc_$NAME.log"; }
done
find "$OUT" -name "*.csv" | head -40
