#!/bin/bash
# Run on the GPU box (via gpurun): every bench line that profiles/<round>/ keeps, one JSON file each
# (gpurun_out/<tag>/lines/bench_n1_*.json).  The headline first; a failure of one line does not stop the rest.
# Usage: tools/bench_all.sh <tag>
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r03}"
OUT="$REPO/gpurun_out/$TAG/lines"
mkdir -p "$OUT"
cd "$REPO"
run() {   # run <file> <bench args...>
  local f="$1"; shift
  python3 bench.py "$@" > "$OUT/$f.json" 2> "$OUT/$f.err"; local rc=$?
  echo "$f rc=$rc $(python3 -c "import json,sys; d=json.load(open('$OUT/$f.json')); print('%.3f M/s  frac %.3f' % (d['value']/1e6, d['roofline']['frac']))" 2>/dev/null)"
}
run bench_n1_ladder --steps 10 --warmup 2
run bench_n1_ladder_secp256k1 --steps 10 --warmup 2 --curve secp256k1
run bench_n1_ladder_ref_compat_p256 --steps 10 --warmup 2 --workload ladder-ref-compat
run bench_n1_ladder_ref_compat_secp256k1 --steps 10 --warmup 2 --workload ladder-ref-compat --curve secp256k1
run bench_n1_ladder_x_only --steps 10 --warmup 2 --workload ladder-x
run bench_n1_ladder_x_only_secp256k1 --steps 10 --warmup 2 --workload ladder-x --curve secp256k1
run bench_n1_windowed_variable_base --steps 10 --warmup 2 --workload windowed
run bench_n1_windowed_variable_base_secp256k1 --steps 10 --warmup 2 --workload windowed --curve secp256k1
run bench_n1_windowed_constant_time --steps 10 --warmup 2 --workload windowed-ct
run bench_n1_windowed_constant_time_secp256k1 --steps 10 --warmup 2 --workload windowed-ct --curve secp256k1
run bench_n1_fixed_base --steps 20 --warmup 2 --workload fixed-base
run bench_n1_fixed_base_secp256k1 --steps 20 --warmup 2 --workload fixed-base --curve secp256k1
run bench_n1_fixed_base_constant_time --steps 20 --warmup 2 --workload fixed-base-ct
run bench_n1_fixed_base_constant_time_secp256k1 --steps 20 --warmup 2 --workload fixed-base-ct --curve secp256k1
run bench_n1_fixed_base_signed7 --steps 20 --warmup 2 --workload fixed-base-signed
run bench_n1_fixed_base_signed7_secp256k1 --steps 20 --warmup 2 --workload fixed-base-signed --curve secp256k1
run bench_n1_fixed_base_big20 --steps 20 --warmup 2 --workload fixed-base-big
# curves registered at run time (round 5): the generic kernels, dense 9-limb prime in SGPRs
run bench_n1_ladder_brainpoolP256r1 --steps 10 --warmup 2 --curve brainpoolP256r1
run bench_n1_ladder_sm2 --steps 10 --warmup 2 --curve sm2
run bench_n1_ladder_frp256v1 --steps 10 --warmup 2 --curve frp256v1
run bench_n1_ladder_radix32_brainpoolP256r1 --steps 4 --warmup 1 --curve brainpoolP256r1 --workload ladder-radix32
run bench_n1_ladder_ref_compat_brainpoolP256r1 --steps 4 --warmup 1 --curve brainpoolP256r1 --workload ladder-ref-compat
run bench_n1_windowed_variable_base_brainpoolP256r1 --steps 10 --warmup 2 --curve brainpoolP256r1 --workload windowed
run bench_n1_windowed_constant_time_brainpoolP256r1 --steps 10 --warmup 2 --curve brainpoolP256r1 --workload windowed-ct
run bench_n1_fixed_base_brainpoolP256r1 --steps 20 --warmup 2 --curve brainpoolP256r1 --workload fixed-base
run bench_n1_fixed_base_constant_time_brainpoolP256r1 --steps 20 --warmup 2 --curve brainpoolP256r1 --workload fixed-base-ct
run bench_n1_fixed_base_signed7_brainpoolP256r1 --steps 20 --warmup 2 --curve brainpoolP256r1 --workload fixed-base-signed
run bench_n1_fixed_base_big20_brainpoolP256r1 --steps 20 --warmup 2 --curve brainpoolP256r1 --workload fixed-base-big
run bench_n1_windowed_variable_base_sm2 --steps 10 --warmup 2 --curve sm2 --workload windowed
run bench_n1_windowed_variable_base_frp256v1 --steps 10 --warmup 2 --curve frp256v1 --workload windowed
run bench_n1_ladder_radix32_p256 --steps 10 --warmup 2 --workload ladder-radix32
run bench_n1_group_mode --steps 10 --warmup 2 --multi group
# the whole N > 1 code path of the one-process-per-GPU mode on the one GPU a builder's box has: RCCL init, side stream, dist.gather
f=bench_n1_nccl_single_rank_rehearsal
ECSIMD_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/$f.json" 2> "$OUT/$f.err"
echo "$f rc=$? $(python3 -c "import json; d=json.load(open('$OUT/$f.json')); print('%.3f M/s, compute only %.3f' % (d['value']/1e6, d['config']['gather']['value_compute_only']/1e6))" 2>/dev/null)"
ls -la "$OUT"
