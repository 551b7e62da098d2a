#!/bin/bash
# tools/gpu_step.sh <name> -- the GPU-box command sequences of a round, one case per step (replaces round 3's tools/dbg/gpu_step_*.sh).
# Usage here: /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_step.sh <name>'; everything is written under gpurun_out/<name>/.
set -o pipefail
name=${1:?step name}; shift
out=gpurun_out/$name; mkdir -p "$out"
export TMPDIR=/tmp
case "$name" in
  r4_group_ct)      # round 4: the RCCL branch on the test double, the k* fix of the constant-time comb, the new VALU rates
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_group or constant_time_fixed_base or exceptional_scalars" > "$out/pytest.txt" 2>&1; rc=$?
    tail -5 "$out/pytest.txt"
    [ $rc -eq 0 ] && timeout -k 10 300 tools/ubench/valu_rates r4 > "$out/valu_rates_r4.txt" 2>&1 && cat "$out/valu_rates_r4.txt"
    exit $rc ;;
  r4_radix)         # round 4: the reduced-radix ZDAU against the oracle, then the A/B of both loop representations
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reduced_radix or scalar_mult_vs_oracle or live_reference" > "$out/pytest.txt" 2>&1; rc=$?
    tail -5 "$out/pytest.txt"
    [ $rc -eq 0 ] && timeout -k 10 600 python tools/radix_ab.py ${1:-22} > "$out/radix_ab.txt" 2>&1; rc=$?; cat "$out/radix_ab.txt"
    exit $rc ;;
  ab)               # A/B of library builds: bash tools/gpu_step.sh ab "<bench args>" name=lib ...
    args=$1; shift
    timeout -k 10 900 python tools/ab_variants.py "$args" "$@" > "$out/ab.txt" 2>&1; rc=$?; cat "$out/ab.txt"; exit $rc ;;
  ab_multi)         # several A/B pairs on one box against ONE other library: bash tools/gpu_step.sh ab_multi <other .so> "<workload> <curve>" ...
    other=$1; shift; : > "$out/ab.txt"
    for spec in "$@"; do
      set -- $spec
      echo "== $1 $2" >> "$out/ab.txt"
      timeout -k 10 300 python tools/ab_variants.py "--workload $1 --curve $2 --global-log2-batch 22 --steps 5 --warmup 1" new=base other=$other >> "$out/ab.txt" 2>&1 || { cat "$out/ab.txt"; exit 1; }
    done
    cat "$out/ab.txt"; exit 0 ;;
  bench)            # the driver's default command, timed by the shell as the driver does
    s0=$(date +%s); timeout -k 10 600 python bench.py "$@" > "$out/bench.json" 2> "$out/bench.err"; rc=$?; s1=$(date +%s)
    echo "rc=$rc wall=$((s1 - s0)) s"; tail -3 "$out/bench.err"; python3 -c "
import json; d=json.load(open('$out/bench.json')); r=d['roofline']; print('value %.3f M/s  ms/step %.2f  frac %.3f (a priori %.3f)' % (d['value']/1e6, d['ms_per_step'], r['frac'], r['frac_of_a_priori_peak'])); print('ref_compat', {k: v for k, v in d.get('ref_compat', {}).items() if k != 'what'}); print('cpu', {k: d['cpu_baseline'][k] for k in ('value','cores','kind','lanes_compared','lanes_differing_from_gpu','lanes_differing_confirmed_by_openssl')})"
    exit $rc ;;
  small_batch)      # round 4: the small-batch route of scalar_mult_base (test), then the latency table against the compiled reference
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_cpp_host_api.py -x -q -m gpu -k "small_base or scalar_mult_vs_oracle or cpp_api or fixtures_from" > "$out/pytest.txt" 2>&1; rc=$?
    tail -5 "$out/pytest.txt"
    [ $rc -eq 0 ] && for c in p256 secp256k1; do timeout -k 10 500 python tools/small_batch.py $c > "$out/small_batch_$c.txt" 2>&1 || rc=$?; cat "$out/small_batch_$c.txt"; done
    exit $rc ;;
  traffic)          # round 4: where the window kernels' bytes come from -- the re-read probe of the calibration, then FETCH_SIZE / WRITE_SIZE / RDREQ_DRAM per kernel
    bash tools/profile_calib.sh r04 > "$out/calib.txt" 2>&1; rc=$?; tail -12 "$out/calib.txt"
    [ $rc -eq 0 ] && TRAFFIC_EXTRA_COUNTERS="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum" bash tools/profile_traffic.sh r04 ${@:-windowed windowed-ct} > "$out/traffic.txt" 2>&1; rc=$?; tail -8 "$out/traffic.txt"
    exit $rc ;;
  soak)             # the ladder (default: radix 29; and REF_SQUARE_COMPAT) against the compiled reference: tools/soak.py <log2 lanes> <batches>
    timeout -k 10 1000 python tools/soak.py ${1:-23} ${2:-4} ${3:-p256,secp256k1} > "$out/soak.txt" 2>&1; rc=$?; tail -12 "$out/soak.txt"; exit $rc ;;
  soak_alg)         # every window kernel (fixed base: 4 combs; variable base: plain, constant-time; x only) against the ladder, lane for lane: tools/soak_windowed.py <log2 lanes> <batches>
    timeout -k 10 1000 python tools/soak_windowed.py ${1:-22} ${2:-32} ${3:-} > "$out/soak_alg.txt" 2>&1; rc=$?; tail -6 "$out/soak_alg.txt"; exit $rc ;;
  bench_all)        # every bench line profiles/rNN keeps
    bash tools/bench_all.sh ${1:-r04} > "$out/bench_all.txt" 2>&1; rc=$?; cat "$out/bench_all.txt" | head -30; exit $rc ;;
  comb29)           # round 4: the combs' additions on 29-bit limbs -- every test that runs a comb, then A/B against the radix-32 build (build/ab_comb32)
    timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fields.py tests/test_openssl_crosscheck.py -x -q -m gpu \
      -k "fixed_base or config3 or comb or exceptional or digit_pattern or small_base or openssl or ecdsa or double_scalar or constant_time or x_coordinate" > "$out/pytest.txt" 2>&1; rc=$?
    tail -5 "$out/pytest.txt"; [ $rc -ne 0 ] && exit $rc
    for w in fixed-base fixed-base-ct fixed-base-signed fixed-base-big; do for c in p256 secp256k1; do
      echo "== $w $c" >> "$out/ab.txt"
      timeout -k 10 300 python tools/ab_variants.py "--workload $w --curve $c --global-log2-batch 22 --steps 10 --warmup 2" radix29=base radix32=build/ab_comb32/libecsimd_hip.so >> "$out/ab.txt" 2>&1 || rc=$?
    done; done
    cat "$out/ab.txt"; exit $rc ;;
  varwin29)         # round 4: the odd-digit window loop on 29-bit limbs -- every test that runs it, then A/B against the radix-32 build (build/ab_vw32)
    timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fields.py tests/test_openssl_crosscheck.py tests/test_gpu_large.py -x -q -m gpu \
      -k "windowed or varwin or digit_pattern or openssl or ecdsa or double_scalar or constant_time or x_coordinate or exceptional or maximum or large or invalid" > "$out/pytest.txt" 2>&1; rc=$?
    tail -5 "$out/pytest.txt"; [ $rc -ne 0 ] && exit $rc
    [ -f build/ab_vw32/libecsimd_hip.so ] || exit 0           # the A/B needs the radix-32 build (-DECS_VARWIN_RADIX=32) beside the tree
    for w in windowed windowed-ct; do for c in p256; do
      echo "== $w $c" >> "$out/ab.txt"
      timeout -k 10 300 python tools/ab_variants.py "--workload $w --curve $c --global-log2-batch 22 --steps 5 --warmup 1" radix29=base radix32=build/ab_vw32/libecsimd_hip.so >> "$out/ab.txt" 2>&1 || rc=$?
    done; done
    cat "$out/ab.txt"; exit $rc ;;
  glv29)            # round 4: the default GLV loop + the checked mixed addition on 29-bit limbs -- the tests that run them, then A/B against build/ab_glv32
    timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fields.py tests/test_openssl_crosscheck.py tests/test_gpu_large.py tests/test_cpp_host_api.py -x -q -m gpu \
      -k "windowed or varwin or digit_pattern or openssl or ecdsa or double_scalar or add or complete or x_coordinate or exceptional or maximum or large or invalid or cpp_api or glv or endomorphism" > "$out/pytest.txt" 2>&1; rc=$?
    tail -5 "$out/pytest.txt"; [ $rc -ne 0 ] && exit $rc
    [ -f build/ab_glv32/libecsimd_hip.so ] || exit 0          # (-DECS_GLV_RADIX=32)
    for w in windowed; do
      echo "== $w secp256k1" >> "$out/ab.txt"
      timeout -k 10 300 python tools/ab_variants.py "--workload $w --curve secp256k1 --global-log2-batch 22 --steps 5 --warmup 1" radix29=base radix32=build/ab_glv32/libecsimd_hip.so >> "$out/ab.txt" 2>&1 || rc=$?
    done
    cat "$out/ab.txt"; exit $rc ;;
  profile)          # rocprofv3 --kernel-trace --stats + separate --pmc passes around bench.py: tools/profile.sh <tag> [bench args] (summaries: tools/summarize_profiles.py)
    tag=${1:-r04_ladder}; shift || true
    bash tools/profile.sh "$tag" "$@" > "$out/profile_$tag.txt" 2>&1; rc=$?; tail -8 "$out/profile_$tag.txt"; exit $rc ;;
  secondary)        # tools/bench_kernels.py -> profiles/rNN/secondary_kernels.{json,txt}
    timeout -k 10 900 python tools/bench_kernels.py > "$out/secondary_kernels.json" 2> "$out/secondary_kernels.txt"; rc=$?; tail -70 "$out/secondary_kernels.txt"; exit $rc ;;
  r5_first)         # round 5: the adapter beside the reference, the scrubbed signing workspace, the shared inversion on small primes, BASE_GENERATOR; then the default bench
    timeout -k 10 900 python -m pytest tests/test_integration_adapter.py tests/test_gpu_fields.py tests/test_gpu_parity.py -x -q -m gpu \
      -k "adapter or ecdsa or prime_flagged or small_base or to_affine or inversion or runtime_modulus or random_odd" > "$out/pytest.txt" 2>&1; rc=$?
    tail -8 "$out/pytest.txt"; [ $rc -ne 0 ] && exit $rc
    ./oracle/_ref/adapter_driver 512 65536 > "$out/adapter_driver.txt" 2>&1; cat "$out/adapter_driver.txt"
    timeout -k 10 600 python bench.py > "$out/bench.json" 2> "$out/bench.err"; rc=$?; tail -3 "$out/bench.err"; head -c 1500 "$out/bench.json"; exit $rc ;;
  r5_curves)        # round 5: curves registered at run time -- the parity tests, then the three loops' rates on brainpoolP256r1 beside P-256's
    timeout -k 10 900 python -m pytest tests/test_gpu_curves.py -x -q -m gpu > "$out/pytest.txt" 2>&1; rc=$?
    tail -12 "$out/pytest.txt"; [ $rc -ne 0 ] && exit $rc
    timeout -k 10 600 python tools/curve_perf.py ${1:-22} > "$out/curve_perf.txt" 2>&1; rc=$?; cat "$out/curve_perf.txt"; exit $rc ;;
  r5_suite)         # round 5: the whole GPU suite, then the bench lines of a registered curve (default loop, canonical words, reference squaring)
    timeout -k 10 1000 python -m pytest tests -x -q -m gpu > "$out/pytest.txt" 2>&1; rc=$?; tail -6 "$out/pytest.txt"; [ $rc -ne 0 ] && exit $rc
    for w in ladder ladder-radix32 ladder-ref-compat; do
      timeout -k 10 300 python bench.py --curve brainpoolP256r1 --workload $w --global-log2-batch 22 --steps 5 --warmup 1 > "$out/bench_brainpool_$w.json" 2> "$out/bench_brainpool_$w.err" || { rc=$?; tail -5 "$out/bench_brainpool_$w.err"; exit $rc; }
      python3 -c "
import json; d=json.load(open('$out/bench_brainpool_$w.json')); r=d['roofline']; c=d['cpu_baseline']; print('$w: %.2f M/s, frac %.3f; cpu %s %.1f k/s on %d cores, %d lanes compared, %d differing' % (d['value']/1e6, r['frac'], c['kind'], c['value']/1e3, c['cores'], c['lanes_compared'], c['lanes_differing_from_gpu']))"
    done; exit 0 ;;
  r5_profile)       # round 5: rocprofv3 kernel stats + --pmc passes for the lines the pipe model speaks about, and the two kernels VERDICT r4 next 7 asks about
    rc=0
    prof() { tag=$1; shift; bash tools/profile.sh "r05_$tag" "$@" > "$out/profile_$tag.txt" 2>&1 || { rc=$?; tail -5 "$out/profile_$tag.txt"; }; echo "profiled $tag rc=$rc"; }
    prof ladder
    prof ladder_secp256k1 --curve secp256k1
    prof ladder_ref_compat --workload ladder-ref-compat
    prof ladder_ref_compat_secp256k1 --workload ladder-ref-compat --curve secp256k1
    prof ladder_brainpoolP256r1 --curve brainpoolP256r1
    prof fixed_base_big --workload fixed-base-big
    prof windowed_ct_secp256k1 --workload windowed-ct --curve secp256k1
    exit $rc ;;
  r5_profile2)      # round 5: the same passes for the remaining ladder lines (the two other registered curves, the canonical-word loops), so that no bench line carries traffic: null
    rc=0
    prof() { tag=$1; shift; bash tools/profile.sh "r05_$tag" "$@" > "$out/profile_$tag.txt" 2>&1 || { rc=$?; tail -5 "$out/profile_$tag.txt"; }; echo "profiled $tag rc=$rc"; }
    prof ladder_sm2 --curve sm2
    prof ladder_frp256v1 --curve frp256v1
    prof ladder_radix32_p256 --workload ladder-radix32
    prof ladder_radix32_brainpoolP256r1 --workload ladder-radix32 --curve brainpoolP256r1
    prof ladder_ref_compat_brainpoolP256r1 --workload ladder-ref-compat --curve brainpoolP256r1
    exit $rc ;;
  r5_small)         # round 5: the small-batch route of a registered curve (tests, latencies), then the generic canonical-word ladders at 2 against 3 waves per SIMD (build/ab_g32w3: -DGLADDER32_WAVES_PER_SIMD=3, 27 / 11 registers spilled)
    timeout -k 10 900 python -m pytest tests/test_gpu_curves.py -x -q -m gpu > "$out/pytest.txt" 2>&1; rc=$?
    tail -5 "$out/pytest.txt"; [ $rc -ne 0 ] && exit $rc
    timeout -k 10 600 python tools/curve_perf.py 22 > "$out/curve_perf.txt" 2>&1 || rc=$?; cat "$out/curve_perf.txt"
    for w in ladder-ref-compat ladder-radix32; do
      echo "== brainpoolP256r1 $w" >> "$out/ab.txt"
      timeout -k 10 400 python tools/ab_variants.py "--curve brainpoolP256r1 --workload $w --global-log2-batch 22 --steps 3 --warmup 1" waves2=base waves3=build/ab_g32w3/libecsimd_hip.so >> "$out/ab.txt" 2>&1 || rc=$?
    done
    cat "$out/ab.txt"; exit $rc ;;
  r5_ab_k1)         # round 5: the secp256k1 Montgomery reduction's rounds on one 64-bit MAC against rounds 1-4's borrow-tracking form (build/ab_k1old, -DECS_K1_REDUCE_MAD64=0)
    timeout -k 10 500 python tools/ab_variants.py "--workload ladder-ref-compat --curve secp256k1 --steps 5 --warmup 1" mad64=base borrow_tracking=build/ab_k1old/libecsimd_hip.so > "$out/ab.txt" 2>&1; rc=$?
    timeout -k 10 300 python tools/ab_variants.py "--workload ladder-ref-compat --curve p256 --steps 5 --warmup 1" p256_for_scale=base >> "$out/ab.txt" 2>&1
    cat "$out/ab.txt"; exit $rc ;;
  r5_adapter)       # round 5: the reference's register layout transposed on the device -- its test, the adapter beside the reference (device transposition, then the host loop: the A/B), the PCIe-inclusive rate
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_integration_adapter.py -x -q -m gpu -k "register_layout or adapter or wire_formats" > "$out/pytest.txt" 2>&1; rc=$?
    tail -5 "$out/pytest.txt"; [ $rc -ne 0 ] && exit $rc
    timeout -k 10 600 ./oracle/_ref/adapter_driver 512 65536 1048576 > "$out/adapter_driver.txt" 2>&1 || rc=$?; cat "$out/adapter_driver.txt"
    ECSIMD_ADAPTER_HOST_TRANSPOSE=1 timeout -k 10 600 ./oracle/_ref/adapter_driver 512 65536 1048576 > "$out/adapter_driver_host_transpose.txt" 2>&1 || rc=$?; cat "$out/adapter_driver_host_transpose.txt"
    timeout -k 10 300 python tools/pcie_rate.py > "$out/pcie_rate.txt" 2>&1 || rc=$?; cat "$out/pcie_rate.txt"
    exit $rc ;;
  r5_final)         # round 5, the final tree on ONE box: the secondary kernels, the registered curves' rates and latencies, the adapter and the PCIe-inclusive rate (bench_all and pytest_gpu are their own steps: time)
    rc=0
    timeout -k 10 900 python tools/bench_kernels.py > "$out/secondary_kernels.json" 2> "$out/secondary_kernels.txt" || rc=$?; tail -12 "$out/secondary_kernels.txt"
    timeout -k 10 600 python tools/curve_perf.py 22 > "$out/curve_perf.txt" 2>&1 || rc=$?; tail -8 "$out/curve_perf.txt"
    timeout -k 10 600 ./oracle/_ref/adapter_driver 512 65536 1048576 > "$out/adapter_driver.txt" 2>&1 || rc=$?; cat "$out/adapter_driver.txt"
    ECSIMD_ADAPTER_HOST_TRANSPOSE=1 timeout -k 10 600 ./oracle/_ref/adapter_driver 512 65536 1048576 > "$out/adapter_driver_host_transpose.txt" 2>&1 || rc=$?; tail -4 "$out/adapter_driver_host_transpose.txt"
    timeout -k 10 300 python tools/pcie_rate.py > "$out/pcie_rate.txt" 2>&1 || rc=$?; cat "$out/pcie_rate.txt"
    exit $rc ;;
  r5_gvarwin)       # round 5: the variable-base window loop of a registered curve (k_gvarwin.hip) -- its tests, then its rates beside the ladder's
    timeout -k 10 600 python -m pytest tests/test_gpu_curves.py tests/test_gpu_witness.py -x -q -m gpu > "$out/pytest.txt" 2>&1; rc=$?
    tail -15 "$out/pytest.txt"; [ $rc -ne 0 ] && exit $rc
    timeout -k 10 500 python tools/gvarwin_perf.py ${1:-22} > "$out/gvarwin_perf.txt" 2>&1; rc=$?; cat "$out/gvarwin_perf.txt"; exit $rc ;;
  r5_gvw_lines)     # round 5: the window loop of a registered curve -- its bench lines (CPU leg: the reference instantiated for the curve + to_affine), rocprofv3 stats + --pmc passes, the soak against the ladder
    rc=0; mkdir -p "$out/lines"
    for c in brainpoolP256r1 sm2 frp256v1; do
      timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --curve $c --workload windowed > "$out/lines/bench_n1_windowed_variable_base_$c.json" 2> "$out/lines/bench_n1_windowed_variable_base_$c.err" || rc=$?
      python3 -c "import json; d=json.load(open('$out/lines/bench_n1_windowed_variable_base_$c.json')); print('$c %.3f M/s  frac %.3f' % (d['value']/1e6, d['roofline']['frac']), {k: v for k, v in d['cpu_baseline'].items() if k.startswith('lanes') or k in ('value', 'kind')})" || { tail -5 "$out/lines/bench_n1_windowed_variable_base_$c.err"; rc=1; }
    done
    [ $rc -ne 0 ] && exit $rc
    bash tools/profile.sh r05_windowed_brainpoolP256r1 --workload windowed --curve brainpoolP256r1 > "$out/profile.txt" 2>&1 || { rc=$?; tail -5 "$out/profile.txt"; }
    echo "profiled rc=$rc"; [ $rc -ne 0 ] && exit $rc
    timeout -k 10 600 python tools/soak_windowed.py 22 ${1:-4} brainpoolP256r1,sm2,frp256v1 > "$out/soak_alg.txt" 2>&1; rc=$?; tail -4 "$out/soak_alg.txt"; exit $rc ;;
  r5_gvw_profile2)  # round 5: counter passes for the window loop's remaining lines (no bench line without its traffic)
    rc=0
    prof() { tag=$1; shift; bash tools/profile.sh "r05_$tag" "$@" > "$out/profile_$tag.txt" 2>&1 || { rc=$?; tail -5 "$out/profile_$tag.txt"; }; echo "profiled $tag rc=$rc"; }
    prof windowed_sm2 --workload windowed --curve sm2
    prof windowed_frp256v1 --workload windowed --curve frp256v1
    prof windowed_ct_brainpoolP256r1 --workload windowed-ct --curve brainpoolP256r1
    exit $rc ;;
  r5_gcomb_profile) # round 5: counter passes for the registered curve's four combs
    rc=0
    prof() { tag=$1; shift; bash tools/profile.sh "r05_$tag" "$@" > "$out/profile_$tag.txt" 2>&1 || { rc=$?; tail -5 "$out/profile_$tag.txt"; }; echo "profiled $tag rc=$rc"; }
    prof fixed_base_brainpoolP256r1 --workload fixed-base --curve brainpoolP256r1
    prof fixed_base_ct_brainpoolP256r1 --workload fixed-base-ct --curve brainpoolP256r1
    prof fixed_base_signed_brainpoolP256r1 --workload fixed-base-signed --curve brainpoolP256r1
    prof fixed_base_big_brainpoolP256r1 --workload fixed-base-big --curve brainpoolP256r1
    exit $rc ;;
  r5_gvw_relines)   # round 5: the window loop's bench lines again, with the committed counter traffic in them
    rc=0; mkdir -p "$out/lines"
    line() { f=$1; shift; timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 "$@" > "$out/lines/$f.json" 2> "$out/lines/$f.err" || rc=$?
      python3 -c "import json; d=json.load(open('$out/lines/$f.json')); print('$f %.3f M/s  frac %.3f traffic %s' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['traffic']))" || { tail -5 "$out/lines/$f.err"; rc=1; }; }
    for c in brainpoolP256r1 sm2 frp256v1; do line bench_n1_windowed_variable_base_$c --curve $c --workload windowed; done
    line bench_n1_windowed_constant_time_brainpoolP256r1 --curve brainpoolP256r1 --workload windowed-ct
    [ "$1" = "fixed" ] || exit $rc
    line bench_n1_fixed_base_brainpoolP256r1 --curve brainpoolP256r1 --workload fixed-base --steps 20
    line bench_n1_fixed_base_constant_time_brainpoolP256r1 --curve brainpoolP256r1 --workload fixed-base-ct --steps 20
    line bench_n1_fixed_base_signed7_brainpoolP256r1 --curve brainpoolP256r1 --workload fixed-base-signed --steps 20
    line bench_n1_fixed_base_big20_brainpoolP256r1 --curve brainpoolP256r1 --workload fixed-base-big --steps 20
    exit $rc ;;
  pytest_gpu)       # the whole GPU suite, as the driver runs it
    timeout -k 10 1100 python -m pytest tests -x -q -m gpu > "$out/pytest.txt" 2>&1; rc=$?; tail -15 "$out/pytest.txt"; exit $rc ;;
  *) echo "unknown step $name"; exit 2 ;;
esac
