#!/bin/bash
# Run on the GPU box (via gpurun): HBM-traffic counters for EVERY bench workload on both curves (no
# `traffic: null`).  One rocprofv3 --pmc pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass), bench.py at its
# default launch size (2^24 lanes), 2 steps, no CPU leg.  Output: gpurun_out/traffic_<tag>/<workload>_<curve>/pmc_<COUNTER>/...;
# tools/summarize_traffic.py turns it into profiles/pmc_traffic.json + profiles/<round>/traffic/*.json.
# Usage: tools/profile_traffic.sh <tag> [workload ...]
set -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r03}"; shift || true
WORKLOADS="${*:-ladder ladder-ref-compat ladder-x windowed fixed-base fixed-base-signed fixed-base-big}"
OUT="$REPO/gpurun_out/traffic_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ECSIMD_BENCH_STEP_MARKER=1      # a one-element k_fill_random launch in front of every timed step
for W in $WORKLOADS; do
  for CV in p256 secp256k1; do
    D="$OUT/${W}_${CV}"; mkdir -p "$D"
    # TRAFFIC_EXTRA_COUNTERS (r4): further single-counter passes, e.g. TCC_EA0_RDREQ_DRAM_sum (the requests "destined for DRAM (MC)")
    for C in FETCH_SIZE WRITE_SIZE $TRAFFIC_EXTRA_COUNTERS; do
      rocprofv3 --pmc $C --output-format csv -d "$D/pmc_$C" -- python3 "$REPO/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --workload $W --curve $CV > "$D/pmc_$C.log" 2>&1 \
        || { echo "FAILED $W $CV $C"; tail -3 "$D/pmc_$C.log"; }
    done
    echo "done $W $CV"
  done
done
