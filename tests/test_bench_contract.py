"""CPU suite: the committed bench lines carry every field the driver's contract names, and the numbers in
them are internally consistent (achieved = value x algorithmic work; frac = achieved / peak)."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_n1_*.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.relpath(p, ROOT) for p in LINES])
def test_bench_line_contract(path):
    d = json.load(open(path))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    per_launch = d["config"]["per_gpu_batch"] / (r["kernel_ms"] * 1e-3) * r["algorithmic_mad32_per_unit"] / 1e12
    assert abs(per_launch - r["achieved"]) / r["achieved"] < 1e-6
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in c, key
        assert c["kind"] in ("reference", "port") and c["differences_all_explained_by_reference_square_defect"] is True
        if "lanes_differing_confirmed_by_openssl" in c and c["lanes_differing_confirmed_by_openssl"] is not None:      # round 2 on
            assert c["lanes_differing_confirmed_by_openssl"] == c["lanes_differing_from_gpu"]
    assert "parity_failures" not in d


def test_there_is_a_headline_line():
    assert any("ladder" in p and "secp" not in p for p in LINES)


def test_bench_defaults_are_config_4_verbatim():
    """BASELINE.json configs[3]: variable base, batch 2^24 over the GPUs of the node (strong scaling)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    a = bench.parse_args([])
    assert (a.gpus, a.scaling, a.global_log2_batch, a.workload, a.curve) == (1, "strong", 24, "ladder", "p256")
    from ecsimd_amd.shard import plan
    assert plan(a.scaling, 1 << a.global_log2_batch, 5, 8) == (5 << 21, 1 << 21, 1 << 24, 1 << 21)


def test_committed_ladder_traffic_is_the_ladder_kernels_own():
    """roofline.traffic comes from profiles/pmc_traffic.json (the counters cannot be read inside the run): the default
    workload's figure must be the ladder kernel's alone -- its algorithmic 192 B per scalar plus a little -- not the sum
    over every kernel of the profiled command."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    n = 1 << 24
    traffic, src = bench.committed_traffic(bench.parse_args([]), n)
    assert traffic is not None and "not measured in this run" in src
    assert 1.0 <= traffic / (bench.ALGO_BYTES_PER_SCALAR_MULT * n) < 1.1


def test_a_wrong_gpu_result_fails_the_benchmark():
    """a kernel regression must not print a headline and exit 0.  bench.py's cpu_baseline leg is driven here with a
    stand-in "GPU" (numpy arrays holding the exact oracle's results): untouched it reports no failure; with ONE corrupted lane the
    two oracles cannot attribute the difference to the reference's square() defect, libcrypto does not confirm it, the leg appends
    to `failures`, and main() turns a non-empty `failures` into exit code EXIT_PARITY (checked on the source)."""
    import importlib.util
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import fill_random_np, SEED, P256
    from oracle import loader
    spec = importlib.util.spec_from_file_location("bench_for_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    ora = loader.Oracle(faithful=False)

    class FakeEngine:                                   # numpy in, numpy out: what bench.py needs from an Engine on this leg
        to_numpy = staticmethod(lambda t: t)
        select_rows = staticmethod(lambda t, rows: t[rows])
        to_affine = staticmethod(lambda curve, j: ora.to_affine(curve, j))

    n = 256 * bench.usable_cores()
    k = fill_random_np(n, SEED, 1); s = fill_random_np(n, SEED, 2)
    c = ora.constants(P256)
    bx, by = ora.to_affine(P256, ora.scalar_mult(P256, s, np.tile(c["gx"], (n, 1)), np.tile(c["gy"], (n, 1)), threads=8))
    good = [a.copy() for a in ora.scalar_mult(P256, k, bx, by, threads=8)]
    failures = []
    res = bench.cpu_baseline(FakeEngine, P256, k, bx, by, good, 0.01, failures)
    assert failures == [] and res["lanes_differing_from_gpu"] == 0 and res["differences_all_explained_by_reference_square_defect"] is True
    bad = [a.copy() for a in good]
    bad[0][5, 0] ^= np.uint64(1)                        # one wrong bit in one X coordinate
    failures = []
    res = bench.cpu_baseline(FakeEngine, P256, k, bx, by, bad, 0.01, failures)
    assert res["lanes_differing_from_gpu"] == 1 and res["differences_all_explained_by_reference_square_defect"] is False
    assert any("square() defect" in f for f in failures)
    if res["lanes_differing_confirmed_by_openssl"] is not None:
        assert res["lanes_differing_confirmed_by_openssl"] == 0 and any("libcrypto" in f for f in failures)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "return EXIT_PARITY if failures else 0" in src and "sys.exit(code)" in src and bench.EXIT_PARITY != 0


# ---------------------------------------------------------------- the N > 1 launcher
STUB = r'''
import json, os, sys
print("some chatter from a rank")
print(json.dumps({"metric": "stub", "argv": sys.argv[1:], "hsa": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "has_rank": "RANK" in os.environ}))
sys.exit(int(os.environ.get("STUB_RC", "0")))
'''


def _run_launcher(tmp_path, extra_env, args):
    import subprocess
    import sys
    stub = tmp_path / "stub_ranks.py"
    stub.write_text(STUB)
    env = dict(os.environ)
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    env.update({"ECSIMD_BENCH_LAUNCHER": f"{sys.executable} {stub}", "ECSIMD_BENCH_LAUNCHER_REPORT": "1"})
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=120)


def test_gpus_n_without_torchrun_launches_the_ranks_as_a_child(tmp_path):
    """`python bench.py --gpus 8` started the way the driver starts the N = 1 run: the parent starts the ranks as a fresh
    child process BEFORE it has touched a GPU (it has loaded neither the HIP runtime nor torch by the time the child has
    finished), relays the flags unchanged, prints the child's one JSON line and nothing else on stdout."""
    r = _run_launcher(tmp_path, {}, ["--gpus", "8", "--steps", "3", "--warmup", "1", "--curve", "secp256k1"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["argv"] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--curve", "secp256k1"]
    assert d["hsa"] == "0"                                   # dmabuf IPC for RCCL (see the environment notes)
    assert "launcher: hip_loaded=False torch_imported=False" in r.stderr
    assert "some chatter from a rank" in r.stderr            # a rank's other output goes to stderr, never between the driver and the line


def test_the_launcher_propagates_a_failing_rank(tmp_path):
    r = _run_launcher(tmp_path, {"STUB_RC": "3"}, ["--gpus", "2"])
    assert r.returncode == 3 and json.loads(r.stdout.strip())["metric"] == "stub"      # EXIT_PARITY after printing survives the relay


def test_the_real_launch_command_starts_and_fails_loudly_without_gpus():
    """No stub: torch.distributed.run really starts two ranks here; without a GPU each dies at its first device call and
    the launcher must come back non-zero with no result line -- never a silent CPU run."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ECSIMD_BENCH_LAUNCHER")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_under_torchrun_bench_does_not_launch_again():
    """With RANK set (the driver's own torch.distributed.run) the process is a rank, not a launcher; and --multi group
    refuses to run as one of several ranks."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if args.gpus > 1 and "RANK" not in os.environ:\n        return launch_ranks(args, sys.argv[1:])' in src
    import subprocess
    import sys
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--multi", "group"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "one process driving every GPU" in r.stderr


def test_config1_ops8_times_the_reference_and_hashes_its_outputs():
    """BASELINE.json configs[0] (benchs/ops.cpp, batch 8, CPU only): the timed loops run inside the checker libraries; the
    compiled reference's outputs hash equal to the restatement's for mgry_sqr_256 / mgry_reduce_512 / mul_256."""
    import importlib.util
    from oracle import loader
    spec = importlib.util.spec_from_file_location("bench_for_test3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    kind = "reference" if loader.reference_available() else "port"
    c1 = bench.config1_ops8(kind)
    assert set(c1["ops"]) == {"mgry_sqr_256", "mgry_reduce_512", "mul_256"} and c1["timed"] == kind
    for name, o in c1["ops"].items():
        assert o["equals_the_restatement"] is True and 1.0 < o["ns_per_wide"] < 1e6, (name, o)
    # the restatement's own results for these inputs, pinned (any change of the oracle's arithmetic shows here)
    assert c1["ops"]["mgry_sqr_256"]["output_sha256_16"] == "92f54302a1cbfc86"
    assert c1["ops"]["mgry_reduce_512"]["output_sha256_16"] == "6dc7617274355fcf"
    assert c1["ops"]["mul_256"]["output_sha256_16"] == "7e57ffaf6954b9b0"


def test_native_chatter_cannot_reach_stdout(tmp_path):
    """RCCL printf()s a version banner on stdout when rank 0 creates its first communicator; the contract is ONE JSON line
    there.  bench.claim_stdout() points file descriptor 1 at stderr and hands back the real stdout."""
    import subprocess
    import sys
    prog = (f"import importlib.util, os, ctypes\n"
            f"spec = importlib.util.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            f"emit = b.claim_stdout()\n"
            f"libc = ctypes.CDLL(None); libc.printf(b'RCCL version : 2.26.6 (a native library talking)\\n'); libc.fflush(None)\n"
            f"print('python chatter')\n"
            f"emit('{{\"metric\": \"x\"}}')\n")
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"metric": "x"}\n'
    assert "RCCL version" in r.stderr and "python chatter" in r.stderr


def test_designs_results_table_is_the_committed_json():
    """VERDICT r2 "record drift": DESIGN.md section 4's table is generated from profiles/r05/bench_n1_*.json."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "results_table.py"), "r05", "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_round3_lines_carry_traffic_and_the_full_cpu_baseline():
    """no `traffic: null`, and cpu_baseline has per_core / one_thread / cpu_model / flags."""
    r3 = [p for p in LINES if os.sep + "r03" + os.sep in p]
    assert len(r3) >= 12
    for path in r3:
        d = json.load(open(path))
        assert d["roofline"]["traffic"] is not None and d["roofline"]["traffic"] > 0, path
        # (the constant-time lines do uncounted work -- reading every table entry -- on top of the additions the algorithmic figure counts)
        assert 0.6 < d["roofline"]["frac_of_a_priori_peak"] < d["roofline"]["frac"] < 0.8 or "compat" in path or ("constant_time" in path and d["roofline"]["frac"] > 0.5), path
        c = d.get("cpu_baseline")
        if c:
            for key in ("per_core", "one_thread", "cpu_model", "flags"):
                assert key in c, (path, key)
            assert c["one_thread"]["seconds"] >= 2.0 and c["one_thread"]["threads"] == 1
            if "config1_ops8" in c:
                assert all(o["equals_the_restatement"] for o in c["config1_ops8"]["ops"].values())


def test_round4_lines_carry_the_reference_compatible_rate_and_the_new_fractions():
    """the default line times ECSIMD_HIP_REF_SQUARE_COMPAT at the same batch and compares it with the compiled reference on the
    CPU leg's lanes -- 0 differing, or the run would have exited with EXIT_PARITY.  Round 4's ladder runs on 29-bit limbs: its algorithmic rate is
    0.78-0.85 of the a-priori multiply peak (rounds 1-3: 0.68); the combs' additions do too."""
    r4 = {os.path.basename(p)[len("bench_n1_"):-len(".json")]: json.load(open(p)) for p in LINES if os.sep + "r04" + os.sep in p}
    assert len(r4) >= 19
    rc = r4["ladder"]["ref_compat"]
    assert rc["lanes_compared"] > 10 ** 6 and rc["lanes_differing"] == 0 and rc["steps"] >= 3
    assert 40e6 < rc["value"] < r4["ladder"]["value"] and 0.55 < rc["frac_of_a_priori_peak"] < 0.7
    assert "variable-base" in r4["ladder"]["config"]["workload"] and "2^24" in r4["ladder"]["config"]["workload"]
    for name in ("ladder", "ladder_secp256k1", "group_mode"):
        assert 54e6 < r4[name]["value"] < 64e6 and 0.75 < r4[name]["roofline"]["frac_of_a_priori_peak"] < 0.9, name
    for name, d in r4.items():
        assert d["roofline"]["traffic"] is not None and d["roofline"]["traffic"] > 0, name
        assert d.get("parity_failures") is None, name
    assert r4["fixed_base"]["value"] > 320e6 and r4["fixed_base_secp256k1"]["value"] > 320e6           # BASELINE configs[2] on the reduced radix (r3: 295)
    assert r4["fixed_base_constant_time"]["value"] > 350e6 and r4["fixed_base_signed7"]["value"] > 530e6


def test_round5_lines_carry_what_the_pipe_does_and_the_registered_curves():
    """every ladder line of round 5 says what the VALU pipe does (instructions per unit, multiplies among them, cycles per instruction
    per SIMD, how much of the elapsed time the instruction mix alone accounts for); the default line's `ref_compat` object is the documented contract (value,
    frac, lanes compared, 0 differing); curves registered at run time have lines of their own, compared with the compiled reference instantiated for them."""
    r5 = {os.path.basename(p)[len("bench_n1_"):-len(".json")]: json.load(open(p)) for p in LINES if os.sep + "r05" + os.sep in p}
    assert len(r5) >= 25
    for name in ("ladder", "ladder_secp256k1", "ladder_ref_compat_p256", "ladder_ref_compat_secp256k1", "ladder_brainpoolP256r1"):
        r = r5[name]["roofline"]
        assert r["multiply_instructions_per_unit"] < r["valu_instructions_per_unit"], name
        assert 3.7 < r["cycles_per_valu_instruction_per_simd"] < 4.3 and 0.9 < r["issue_bound_frac"] < 1.15, name
    rc = r5["ladder"]["ref_compat"]
    assert rc["lanes_compared"] > 10 ** 6 and rc["lanes_differing"] == 0 and rc["steps"] >= 3 and 0.6 < rc["frac"] < 0.75 and 40e6 < rc["value"] < r5["ladder"]["value"]
    for name in ("ladder_brainpoolP256r1", "ladder_sm2", "ladder_frp256v1"):
        d = r5[name]
        assert name[len("ladder_"):] in d["metric"] and 38e6 < d["value"] < 48e6 and 0.6 < d["roofline"]["frac"] < 0.75, name
        c = d["cpu_baseline"]
        assert c["kind"] == "reference" and c["lanes_compared"] > 10 ** 5 and c["lanes_differing_from_gpu"] <= 12, name      # the reference's dropped carry: ~4e-6 of lanes
        # ... every one of them settled for the GPU by textbook affine arithmetic on Python integers (libcrypto's harness has no such curve)
        assert c["lanes_differing_confirmed_by_textbook_arithmetic"] == c["lanes_differing_from_gpu"], name
    for name, d in r5.items():
        assert d.get("parity_failures") is None, name
        assert d["roofline"]["traffic"] is not None and d["roofline"]["traffic"] > 0, name           # a committed counter pass for every line
    for name in ("ladder", "ladder_secp256k1", "group_mode"):
        assert 54e6 < r5[name]["value"] < 64e6 and 0.86 < r5[name]["roofline"]["frac"] < 1.0, name
    assert r5["ladder_ref_compat_secp256k1"]["value"] > 41e6                                                        # r4: 40.06 (the Montgomery rounds on one 64-bit MAC)


def test_the_group_leg_child_is_not_a_rank():
    """At N > 1 rank 0 starts `bench.py --multi group` in a child process; the child must not inherit RANK / WORLD_SIZE (it
    would refuse to run as "one of several ranks").  Without GPUs here it has to get as far as counting devices."""
    import importlib.util
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without GPUs")
    spec = importlib.util.spec_from_file_location("bench_for_test4", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    old = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    os.environ.update({"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    try:
        res = bench.group_leg(bench.parse_args(["--gpus", "2", "--steps", "2"]), 2, timeout_s=120)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert res["ok"] is False and "device(s) visible" in res["stderr_tail"] and "torch.distributed.run" not in res["stderr_tail"], res


# ---------------------------------------------------------------- round 5: the line says what the pipe does, and the committed figures cannot go stale silently
def test_the_committed_pipe_figures_describe_the_isa_the_build_ships():
    """roofline.{valu_instructions_per_unit, multiply_instructions_per_unit, cycles_per_valu_instruction_per_simd, issue_bound_frac}
    are read from profiles/pmc_pipe.json (counters of committed rocprofv3 --pmc passes + the instruction mix of the shipped ISA, tools/pipe_model.py).  The
    day a kernel changes without a re-profile, the loop mix recorded there differs from the ISA `make` leaves under build/csrc: this test fails."""
    import subprocess
    import sys
    if not os.path.exists(os.path.join(ROOT, "build", "csrc", "k_ladder_p256-hip-amdgcn-amd-amdhsa-gfx950.s")):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pipe_model.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    pipe = json.load(open(os.path.join(ROOT, "profiles", "pmc_pipe.json")))["kernels"]
    lad = pipe["k_scalar_mult_p256_2^24"]
    # the static count (loop mix x 254 + the straight-line rest) and the SQ_INSTS_VALU counter agree: the ISA the counters saw IS this one
    assert abs(lad["static_valu_instructions_per_unit"] - lad["valu_instructions_per_unit"]) / lad["valu_instructions_per_unit"] < 0.002
    assert 3.8 < lad["cycles_per_valu_instruction_per_simd"] < 4.3 and 0.9 < lad["issue_bound_frac"] <= 1.02
    assert lad["loop_multiply_instructions"] == 1620 and lad["multiply_instructions_per_unit"] < 555968          # 101 multiply instructions per field multiplication, not 136


def test_the_committed_traffic_is_the_newest_profiles():
    """... and roofline.traffic (profiles/pmc_traffic.json) for the headline equals the HBM bytes of the ladder kernel in the NEWEST profiles/rNN/ladder/pmc_summary.json."""
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "ladder", "pmc_summary.json")))[-1]
    ks = json.load(open(newest))["kernels"]
    k = next(v for name, v in ks.items() if name.startswith("k_scalar_mult<29> @ 16777216"))
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert abs(table["k_scalar_mult_p256_2^24"] - k["hbm_bytes_per_launch"]["total"]) / k["hbm_bytes_per_launch"]["total"] < 0.002, newest
    pipe = json.load(open(os.path.join(ROOT, "profiles", "pmc_pipe.json")))["kernels"]["k_scalar_mult_p256_2^24"]
    assert pipe["source"] == os.path.relpath(newest, ROOT)


def test_ref_compat_is_part_of_the_documented_contract():
    """`ref_compat` -- how fast the path is that is IDENTICAL to the reference on every lane -- is a first-class object of the default line
    (README "The bench line"), beside `value` -- how fast the path is that is RIGHT.  Every committed default-workload line since round 4 carries it whole."""
    readme = open(os.path.join(ROOT, "README.md")).read()
    for word in ("ref_compat", "lanes_compared", "lanes_differing", "issue_bound_frac", "cycles_per_valu_instruction_per_simd"):
        assert word in readme, word
    lines = [p for p in LINES if os.path.basename(p) in ("bench_n1_ladder.json", "bench_n1_ladder_secp256k1.json") and os.sep + "r0" in p and int(p.split(os.sep + "r0")[1][0]) >= 4]
    assert len(lines) >= 2
    for p in lines:
        rc = json.load(open(p))["ref_compat"]
        for key in ("value", "unit", "kernel_ms", "steps", "frac", "frac_of_a_priori_peak", "lanes_compared", "lanes_differing", "compared_with"):
            assert key in rc, (p, key)
        assert rc["lanes_differing"] == 0 and rc["lanes_compared"] > 10 ** 6 and rc["unit"] == "scalar_mults/s"
