#!/bin/bash
# Run on the GPU box (via gpurun): builds tools/ubench/hbm_gather_calib.hip, runs it plainly (GB/s per access pattern) and
# under rocprofv3 --pmc, one pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Output under
# gpurun_out/calib_<tag>/; tools/summarize_calib.py turns it into profiles/<round>/hbm_counter_calibration.json.
set -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r03}"
OUT="$REPO/gpurun_out/calib_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_gather_calib "$REPO/tools/ubench/hbm_gather_calib.hip" 2> "$OUT/build.log" || { cat "$OUT/build.log"; exit 1; }
/tmp/hbm_gather_calib > "$OUT/plain.jsonl" 2> "$OUT/plain.err" || { cat "$OUT/plain.err"; exit 1; }
cat "$OUT/plain.jsonl"
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_RD_[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*" | sort -u > "$OUT/tcc_counters_available.txt"
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum"; do
  NAME=$(echo "$C" | tr ' ' '_')
  echo "== pmc $C"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -- /tmp/hbm_gather_calib > "$OUT/pmc_$NAME.log" 2>&1 || { tail -5 "$OUT/pmc_$NAME.log"; }
done
find "$OUT" -name "*counter_collection.csv" | head
