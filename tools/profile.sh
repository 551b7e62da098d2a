#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# Usage: tools/profile.sh <tag>        outputs under gpurun_out/prof_<tag>/
set -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r01}"
shift || true
EXTRA="$*"          # extra bench.py arguments, e.g. --workload fixed-base
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $EXTRA"
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$BENCH" $ARGS > "$OUT/stats.log" 2>&1 || { tail -20 "$OUT/stats.log"; exit 1; }
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  NAME=$(echo "$C" | tr ' ' '_' | cut -c1-40)
  echo "== pmc $C"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -- python3 "$BENCH" --steps 2 --warmup 0 --no-cpu-baseline $EXTRA > "$OUT/pmc_$NAME.log" 2>&1 || { tail -5 "$OUT/pmc_$NAME.log"; }
done
find "$OUT" -name "*.csv" | head -40
