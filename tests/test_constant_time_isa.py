"""CPU suite: the constant-time claim of INTEGRATION.md ("timing") is checked on the ISA the build ships.

The ladder translation units are compiled to gfx950 assembly with the Makefile's own flags and tools/ct_check.py walks the
254-iteration bit loop of k_scalar_mult (both curves) and k_scalar_mult_x: every branch must hang on a scalar compare of the
iteration counter, no lane mask or VALU result may reach SCC / a branch / an address, the one global load (the scalar-word
reload kwords[nb >> 5], point.cuh ladder_core) must be addressed by loop-invariant registers and the counter, and nothing may
move a field word to the scalar unit.  The same checker must REFUSE the build variant DESIGN.md section 3 rejected for this
very reason (-DECS_COND_SUB=2: a wave-uniform branch on r[7] == 0xffffffff)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ecsimd_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ct_check  # noqa: E402


def shipped_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^HIPFLAGS\s*\?=\s*(.*)$", mk, re.M).group(1)
    arch = re.search(r"^ARCH\s*\?=\s*(\S+)", mk, re.M).group(1)
    return flags.replace("$(ARCH)", arch).split()


_made = []


def assembly(tmp_path_factory, unit, extra=()):
    """The gfx950 listing of one translation unit.  Without extra flags: the listing the Makefile left beside the object it linked into libecsimd_hip.so
    (-save-temps=obj; `make` first, a no-op when the tree is built) -- the ISA that SHIPS, and no second compilation (a third of the CPU suite's time until
    round 5).  With extra flags (the refused build variants): compiled here with the Makefile's own flags."""
    if not extra:
        if not _made:
            subprocess.run(["make", "-j", str(min(8, os.cpu_count() or 1)), "-C", CSRC, "ARCH=gfx950"], check=True, capture_output=True, timeout=1800)
            _made.append(True)
        listing = os.path.join(ROOT, "build", "csrc", unit + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        assert os.path.exists(listing), "the Makefile no longer leaves the device listings in build/csrc (-save-temps=obj)"
        assert os.path.getmtime(listing) >= os.path.getmtime(os.path.join(CSRC, unit + ".hip")), listing
        return open(listing).read()
    out = tmp_path_factory.mktemp("isa") / (unit + "".join(extra).replace("=", "_").replace("-", "") + ".s")
    cmd = ["hipcc"] + [f for f in shipped_flags() if f != "-fPIC"] + list(extra) + ["-S", "--cuda-device-only", os.path.join(CSRC, unit + ".hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, cwd=CSRC, timeout=900)
    return out.read_text()


@pytest.fixture(scope="module")
def p256_asm(tmp_path_factory):
    return assembly(tmp_path_factory, "k_ladder_p256")


@pytest.fixture(scope="module")
def secp256k1_asm(tmp_path_factory):
    return assembly(tmp_path_factory, "k_ladder_secp256k1")


def test_the_shipped_build_uses_the_branch_free_conditional_subtraction():
    src = open(os.path.join(CSRC, "field.cuh")).read()
    assert re.search(r"#ifndef ECS_COND_SUB\s*\n#define ECS_COND_SUB 1\s*\n#endif", src)
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "ECS_COND_SUB" not in mk and "ECS_COND_SUB" not in open(os.path.join(ROOT, "__graft_entry__.py")).read()


# k_scalar_mult<29>: the reduced-radix loop every default call runs since round 4 (fe29.cuh: multiply-adds into carry-free columns, limb-wise adds, shifts
# and masks -- no compare at all); k_scalar_mult<32>: the 8-word loop behind ECSIMD_HIP_LADDER_RADIX32 / REF_SQUARE_COMPAT; k_scalar_mult_x: the ladder without Z (on 29-bit limbs too, later in round 4)
@pytest.mark.parametrize("kernel", ["13k_scalar_multILi29E", "13k_scalar_multILi32E", "15k_scalar_mult_xE"])
def test_p256_bit_loop_is_constant_time(p256_asm, kernel):
    rep = ct_check.check(p256_asm, kernel)
    assert rep["instructions"] > (2200 if "mult_x" in kernel else 2500)    # it IS the ZDAU loop, not some small one (without Z on 29-bit limbs: 2 426)
    # the loop's branches: the exit test on the bit counter and the test(s) that gate the scalar-word reload (one or two, as the compiler
    # arranges `(nb & 31) == 0` and `nb < 256`) -- nothing else, and every one a scalar compare of the counter with a constant
    assert len(rep["branches"]) in (2, 3) and all(re.match(r"s_cmpk?_(lg|eq)_[iu]32 s\d+, (0x101|0x100|0) ; s_cbranch_scc[01] ", b) for b in rep["branches"]), rep["branches"]
    assert len(rep["global_loads"]) == 1 and rep["global_loads"][0].startswith("global_load_dword ") and rep["scratch"] == 0


@pytest.mark.parametrize("kernel", ["13k_scalar_multILi29E", "13k_scalar_multILi32E"])
def test_secp256k1_bit_loop_is_constant_time(secp256k1_asm, kernel):
    rep = ct_check.check(secp256k1_asm, kernel)
    assert rep["instructions"] > 2500 and len(rep["branches"]) in (2, 3) and len(rep["global_loads"]) == 1
    # register spills, if the allocation has any, are constant-address scratch accesses (checked inside ct_check)


@pytest.mark.parametrize("unit", ["k_ladder_p256_refsqr", "k_ladder_secp256k1_refsqr"])
def test_reference_compatible_ladders_are_constant_time_too(tmp_path_factory, unit):
    """ECSIMD_HIP_REF_SQUARE_COMPAT swaps the squaring (field.cuh sqr8_ref: lane-mask carries, no compare, no branch): the header
    calls it safe for secret scalars, so its bit loop passes the same checks."""
    rep = ct_check.check(assembly(tmp_path_factory, unit), "13k_scalar_multILi32E")
    assert rep["instructions"] > 3500 and len(rep["branches"]) == 3 and len(rep["global_loads"]) == 1


def test_the_checker_refuses_the_data_dependent_variant(tmp_path_factory):
    """-DECS_COND_SUB=2 (field.cuh cond_sub_p_guard) branches on `r[7] == 0xffffffff`: the checker has to see it."""
    bad = assembly(tmp_path_factory, "k_ladder_p256", extra=("-DECS_COND_SUB=2",))
    with pytest.raises(ct_check.Violation):
        ct_check.check(bad, "13k_scalar_multILi32E")


def test_the_checker_refuses_planted_leaks(p256_asm):
    """Mutation checks on the real assembly: a lane-mask branch, a field word moved to the scalar unit, and a reload whose
    address was touched by a data register must each be caught."""
    lines = p256_asm.splitlines()
    fn = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*13k_scalar_multILi29E\w*:", ln))       # inside THIS kernel (the unit holds five)
    at = next(i for i, ln in enumerate(lines) if i > fn and "s_cmpk_lg_i32" in ln and "0x101" in ln)
    for planted in ("\ts_cbranch_vccnz .LBB0_7", "\tv_readfirstlane_b32 s40, v20", "\tv_cmp_eq_u32_e64 s[40:41], v20, v21\n\ts_cmp_lg_u64 s[40:41], 0\n\ts_cbranch_scc1 .LBB0_7"):
        mutated = "\n".join(lines[:at] + planted.split("\n") + lines[at:])
        with pytest.raises(ct_check.Violation):
            ct_check.check(mutated, "13k_scalar_multILi29E")
    # the scalar-word reload: in the rotated loop of k_scalar_mult<29> it sits in the blocks in FRONT of the annotated header
    ld = max(i for i, ln in enumerate(lines) if fn < i < at and re.match(r"\s*global_load_dword v\d+, v\[", ln))
    addr = re.search(r"global_load_dword v\d+, v\[(\d+):\d+\]", lines[ld]).group(1)
    mutated = "\n".join(lines[:ld] + [f"\tv_add_u32_e32 v{addr}, v{addr}, v20"] + lines[ld:])
    with pytest.raises(ct_check.Violation):
        ct_check.check(mutated, "13k_scalar_multILi29E")


# ---- the constant-time fixed-base kernel (ECSIMD_HIP_ALG_CONSTANT_TIME: scalar_mult_base + ALG_WINDOWED)
@pytest.fixture(scope="module", params=["k_affine_p256", "k_affine_secp256k1"])
def affine_asm(request, tmp_path_factory):
    return assembly(tmp_path_factory, request.param)


def test_constant_time_fixed_base_kernel_reads_every_entry_and_branches_on_nothing(affine_asm):
    """k_base_windowed<true>: the window loop has ONE branch (the window counter), no global access, and its 32 LDS reads -- all 8
    entries of the window -- are addressed by loop-invariant registers and the counter; around the loop (first window, the
    exceptional scalar k*, k = 0 mod n) nothing branches on a lane mask once the scalar has been loaded."""
    rep = ct_check.check(affine_asm, "15k_base_windowedILb1E", allow_global_loads=0, allow_lds_reads=True)
    assert rep["instructions"] > 1800 and rep["lds_reads"] == 32 and rep["scratch"] == 0 and rep["global_loads"] == []
    assert len(rep["branches"]) == 1 and re.match(r"s_cmpk?_(lg|eq)_[iu]32 s\d+, \S+ ; s_cbranch_scc[01] ", rep["branches"][0]), rep["branches"]
    assert ct_check.check_after_secret_load(affine_asm, "15k_base_windowedILb1E") > 2500


def test_constant_time_five_bit_comb(affine_asm):
    """What ALG_CONSTANT_TIME runs on a fixed base, both curves: k_base_windowed_s<5, true, 256> -- 16 entries x 4 reads per window, one branch,
    nothing else; the same template without the flag (a digit addresses the one read) is refused."""
    rep = ct_check.check(affine_asm, "k_base_windowed_sILi5ELb1ELi256E", allow_global_loads=0, allow_lds_reads=True)
    assert rep["instructions"] > 1900 and rep["lds_reads"] == 64 and rep["scratch"] == 0 and rep["global_loads"] == []
    assert len(rep["branches"]) == 1 and re.match(r"s_cmpk?_(lg|eq|lt|gt|le|ge)_[iu]32 s\d+, \S+ ; s_cbranch_scc[01] ", rep["branches"][0]), rep["branches"]
    assert ct_check.check_after_secret_load(affine_asm, "k_base_windowed_sILi5ELb1ELi256E") > 2800
    with pytest.raises(ct_check.Violation, match="LDS address"):
        ct_check.check(affine_asm, "k_base_windowed_sILi7ELb0E", allow_global_loads=0, allow_lds_reads=True)


def test_the_checker_refuses_the_table_lookup_by_digit(affine_asm):
    """The default kernel reads ONE entry per window at an address made of scalar digits: the same check must refuse it."""
    with pytest.raises(ct_check.Violation, match="LDS address"):
        ct_check.check(affine_asm, "15k_base_windowedILb0E", allow_global_loads=0, allow_lds_reads=True)


@pytest.mark.parametrize("unit", ["k_varwin_p256", "k_varwin_secp256k1"])
def test_constant_time_variable_base_window_loop(tmp_path_factory, unit):
    """k_varwin_mult_odd<true> (ALG_WINDOWED | ALG_CONSTANT_TIME on a variable base): per window 32 global loads -- a quarter of each of the
    lane's 8 table entries, two whole entries (one 128-byte line) at a time -- addressed by loop-invariant registers (the lane's own block), one branch on the window counter,
    nothing else; and the default loop, whose one entry per window is addressed by a digit, must be refused."""
    asm = assembly(tmp_path_factory, unit)
    rep = ct_check.check(asm, "k_varwin_mult_oddILb1E", allow_global_loads=32)
    assert rep["instructions"] > 5000 and len(rep["global_loads"]) == 32          # (a few spills: scratch accesses are checked for constant addressing)
    assert all(g.startswith("global_load_dwordx4 ") for g in rep["global_loads"])
    assert len(rep["branches"]) == 1 and re.match(r"s_cmpk?_(lg|eq|lt|gt|le|ge)_[iu]32 s\d+, \S+ ; s_cbranch_scc[01] ", rep["branches"][0]), rep["branches"]
    assert rep.get("long_jumps", 0) <= 1                       # the back edge of a 78 KB loop body is a relaxed branch
    assert ct_check.check_after_secret_load(asm, "k_varwin_mult_oddILb1E") > 5000
    with pytest.raises(ct_check.Violation, match="address"):
        ct_check.check(asm, "k_varwin_mult_oddILb0E", allow_global_loads=32)


def test_constant_time_glv_loop_of_secp256k1(tmp_path_factory):
    """k_varwin_mult_glv_ct (ALG_WINDOWED | ALG_CONSTANT_TIME on secp256k1): the GLV split on the complete addition law.  Per window 32 global
    loads (the lane's 8 entries, a line at a time) addressed by loop-invariant registers; every branch hangs on the window counter (the exit
    test and the top window's "no doublings yet"); nothing else -- and the default GLV loop (digit-addressed reads, add_checked's branch on
    the operands) must be refused."""
    asm = assembly(tmp_path_factory, "k_varwin_secp256k1")
    rep = ct_check.check(asm, "k_varwin_mult_glv_ctILi0E", allow_global_loads=32)
    assert rep["instructions"] > 8000 and len(rep["global_loads"]) == 32
    assert 1 <= len(rep["branches"]) <= 3 and all(re.match(r"s_cmpk?_(lg|eq|lt|gt|le|ge)_[iu]32 s\d+, \S+ ; s_cbranch_scc[01] ", b) for b in rep["branches"]), rep["branches"]
    assert ct_check.check_after_secret_load(asm, "k_varwin_mult_glv_ctILi0E") > 8000
    with pytest.raises(ct_check.Violation):
        ct_check.check(asm, "17k_varwin_mult_glvILi4E", allow_global_loads=32)


# ---- secret-flow (taint) analysis: the signing path
# ct_check.check_secret_flow follows every value loaded through a pointer argument named secret to the places where it could change what the
# machine DOES: branch conditions, addresses, and the lane mask in force at a memory access.  ecdsa_sign = the constant-time comb (k secret),
# k_to_affine_batched on the Jacobian k G (X, Y, Z as secret as the nonce), k_ecdsa_sign_scalars (d, k, the prefix products in s[]).
@pytest.fixture(scope="module")
def gfield_asm(tmp_path_factory):
    return assembly(tmp_path_factory, "k_gfield")


def test_signing_scalar_field_kernel_keeps_d_and_k_out_of_control_flow_and_addresses(gfield_asm):
    # arguments: (gmod, e, d, k, x, r, s, ok, n, lanes, m) -- d, k and s[] (which carries the nonces' prefix products on the way up) are secret
    rep = ct_check.check_secret_flow(gfield_asm, "k_ecdsa_sign_scalars", secret_args=[2, 3, 6])
    assert rep["secret_loads"] >= 6 and rep["instructions"] > 4000 and not rep["secret_scratch"] and not rep["secret_lds"]
    assert rep["public_branches"] > 0          # it HAS lane-dependent control flow (the batch's ragged tail): on public indices only


def test_generic_division_step_inversion_with_a_secret_operand(gfield_asm):
    rep = ct_check.check_secret_flow(gfield_asm, "k_g_inverse_divsteps", secret_args=[1])
    assert rep["secret_loads"] == 2 and rep["instructions"] > 1500
    rep = ct_check.check_secret_flow(gfield_asm, "k_g_inverse_batched", secret_args=[1, 2])       # out[] holds the prefix products
    assert rep["secret_loads"] >= 4


def test_signing_point_kernels_under_the_same_analysis(affine_asm):
    """The comb with a secret k (the loop is also held to the shape check above) and the simultaneous inversion of k G: zero handling by selects."""
    rep = ct_check.check_secret_flow(affine_asm, "k_base_windowed_sILi5ELb1ELi256E", secret_args=[0, 2, 3, 4])
    assert rep["secret_loads"] == 2 and not rep["secret_lds"]          # the table in LDS is public; the scalar never reaches an LDS address
    rep = ct_check.check_secret_flow(affine_asm, "k_to_affine_batchedILb1E", secret_args=[0, 1, 2, 3, 4])
    assert rep["secret_loads"] >= 6
    rep = ct_check.check_secret_flow(affine_asm, "17k_inverse_batched", secret_args=[0, 1])
    assert rep["secret_loads"] >= 4
    # the public-scalar comb addresses LDS by digits of k: the analysis has to say so
    with pytest.raises(ct_check.Violation, match="LDS address"):
        ct_check.check_secret_flow(affine_asm, "k_base_windowed_sILi7ELb0E", secret_args=[0])


def test_the_secret_flow_analysis_refuses_planted_leaks(gfield_asm):
    """Mutations of the real assembly of k_ecdsa_sign_scalars, each placed right after the first load through the secret pointer k: a branch on a
    compare of the loaded word, the word moved to the scalar unit and branched on, a load addressed by it, a load under a lane mask made of it --
    and the controls: the same mutations fed from a PUBLIC register pass."""
    lines = gfield_asm.splitlines()
    fn = next(i for i, ln in enumerate(lines) if re.match(r"^_ZN\w*20k_ecdsa_sign_scalars\w*:", ln))
    end = next(i for i in range(fn, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = ct_check.parse_function(gfield_asm, "k_ecdsa_sign_scalars")
    label = next(b[0] for b in blocks if b[0].startswith(".LBB"))
    # the first global load whose data the analysis calls secret: find it by running the analysis with a tracer on a copy
    secret_reg = None
    for i in range(fn, end):
        m = re.match(r"\s*global_load_dwordx4 v\[(\d+):\d+\], v\[\d+:\d+\], off", lines[i])
        if m:
            probe = "\n".join(lines[:i + 1] + ["\ts_waitcnt vmcnt(0)", f"\tv_readfirstlane_b32 s90, v{m.group(1)}", "\ts_cmp_lg_u32 s90, 0", f"\ts_cbranch_scc1 {label}"] + lines[i + 1:])
            try:
                ct_check.check_secret_flow(probe, "k_ecdsa_sign_scalars", secret_args=[2, 3, 6])
            except ct_check.Violation:
                secret_reg, at = int(m.group(1)), i + 1
                break
    assert secret_reg is not None, "no load the analysis treats as secret"
    v = f"v{secret_reg}"
    plants = {
        "lane-mask branch": [f"\tv_cmp_ne_u32_e32 vcc, 0, {v}", f"\ts_cbranch_vccnz {label}"],
        "scalar branch": [f"\tv_readfirstlane_b32 s90, {v}", "\ts_cmp_lg_u32 s90, 0", f"\ts_cbranch_scc1 {label}"],
        "address": [f"\tv_mov_b32_e32 v250, {v}", "\tv_mov_b32_e32 v251, 0", "\tglobal_load_dword v252, v[250:251], off"],
        "masked load": [f"\tv_cmp_ne_u32_e32 vcc, 0, {v}", "\ts_and_saveexec_b64 s[90:91], vcc", "\tglobal_load_dword v252, v[0:1], off", "\ts_mov_b64 exec, s[90:91]"],
        "exec branch": [f"\tv_cmp_ne_u32_e32 vcc, 0, {v}", "\ts_and_saveexec_b64 s[90:91], vcc", f"\ts_cbranch_execz {label}", "\ts_mov_b64 exec, s[90:91]"],
    }
    for name, planted in plants.items():
        mutated = "\n".join(lines[:at] + planted + lines[at:])
        with pytest.raises(ct_check.Violation):
            ct_check.check_secret_flow(mutated, "k_ecdsa_sign_scalars", secret_args=[2, 3, 6])
        # the control: fed from a public register (the lane's element index, v0 on entry is the work-item id) nothing is reported ...
        public = [p.replace(v + "\n", "v0\n").replace(f", {v}", ", v0") for p in planted]
        # ... as long as the plant does not alter control flow for real (the branches jump to an existing label: harmless to the analysis)
        ct_check.check_secret_flow("\n".join(lines[:fn + 1] + public + lines[fn + 1:]), "k_ecdsa_sign_scalars", secret_args=[2, 3, 6])
    # a conditional MOVE under a secret lane mask is what field.cuh does on purpose: allowed
    ok = [f"\tv_cmp_ne_u32_e32 vcc, 0, {v}", "\ts_and_saveexec_b64 s[90:91], vcc", "\tv_mov_b32_e32 v252, 0", "\ts_mov_b64 exec, s[90:91]"]
    ct_check.check_secret_flow("\n".join(lines[:at] + ok + lines[at:]), "k_ecdsa_sign_scalars", secret_args=[2, 3, 6])


# ---- the ladder of a curve registered at run time (round 5): the same two checks as the built-in ladders -- ECDSA signing on such a curve puts a nonce through it
@pytest.fixture(scope="module")
def gladder_asm(tmp_path_factory):
    return assembly(tmp_path_factory, "k_gladder")


@pytest.mark.parametrize("kernel", ["16k_gc_scalar_multILi29ELb0E", "16k_gc_scalar_multILi32ELb0E", "16k_gc_scalar_multILi32ELb1E"])
def test_registered_curve_ladder_is_constant_time(gladder_asm, kernel):
    rep = ct_check.check(gladder_asm, kernel)
    assert rep["instructions"] > 3400 and len(rep["branches"]) in (2, 3) and len(rep["global_loads"]) == 1
    assert all(re.match(r"s_cmpk?_(lg|eq)_[iu]32 s\d+, (0x101|0x100|0) ; s_cbranch_scc[01] ", b) for b in rep["branches"]), rep["branches"]
    flow = ct_check.check_secret_flow(gladder_asm, kernel, secret_args=[1])            # (gcurve, k, ...): the scalar
    assert flow["secret_loads"] >= 1 and not flow["secret_lds"]


def test_registered_curve_signing_helpers_under_the_secret_flow_analysis(tmp_path_factory):
    asm = assembly(tmp_path_factory, "k_gcurve")
    assert ct_check.check_secret_flow(asm, "k_gc_ladder_safe_scalars", secret_args=[1, 2, 3])["secret_loads"] >= 1     # the nonce, its adjusted copy, the negation flag
    assert ct_check.check_secret_flow(asm, "k_gc_to_affine_batched", secret_args=[1, 2, 3, 4, 5])["secret_loads"] >= 6   # the Jacobian k G


def test_registered_curve_constant_time_comb(tmp_path_factory):
    """k_gc_base_windowed<true> (ecdsa_sign and scalar_mult_base(ALG_WINDOWED | ALG_CONSTANT_TIME) on a registered curve): the built-in constant-time comb's
    three checks -- one branch in the window loop (its counter), 32 LDS reads at loop-invariant addresses, no global access; nothing branches on a lane
    mask after the scalar is loaded; no secret (the scalar, the Jacobian k G) reaches an address, a branch, EXEC at a memory access or a lane-crossing
    instruction -- and the same template without the flag is refused by both analyses."""
    asm = assembly(tmp_path_factory, "k_gcomb")
    kern = "18k_gc_base_windowedILb1E"
    rep = ct_check.check(asm, kern, allow_global_loads=0, allow_lds_reads=True)
    assert rep["instructions"] > 2400 and rep["lds_reads"] == 32 and rep["scratch"] == 0 and rep["global_loads"] == []
    assert len(rep["branches"]) == 1 and re.match(r"s_cmpk?_(lg|eq)_[iu]32 s\d+, \S+ ; s_cbranch_scc[01] ", rep["branches"][0]), rep["branches"]
    assert ct_check.check_after_secret_load(asm, kern) > 4000
    flow = ct_check.check_secret_flow(asm, kern, secret_args=[2, 4, 5, 6])             # (gcurve, order, k, table, ox, oy, oz, n)
    assert flow["secret_loads"] >= 1 and not flow["secret_lds"] and not flow["secret_scratch"]
    with pytest.raises(ct_check.Violation, match="LDS address"):
        ct_check.check(asm, "18k_gc_base_windowedILb0E", allow_global_loads=0, allow_lds_reads=True)
    with pytest.raises(ct_check.Violation, match="LDS address"):
        ct_check.check_secret_flow(asm, "18k_gc_base_windowedILb0E", secret_args=[2])


def test_registered_curve_constant_time_five_bit_comb(tmp_path_factory):
    """k_gc_base_windowed_s<5, true, 256> (scalar_mult_base(ALG_WINDOWED | ALG_CONSTANT_TIME) and k G of ecdsa_sign on a registered curve since the second
    session of round 5): one branch in the window loop (its counter), 64 LDS reads per window (16 entries x 64 B) at loop-invariant addresses, no global
    access in the loop; no secret (the scalar, the Jacobian k G) reaches an address, a branch, EXEC at a memory access or a lane-crossing instruction -- and
    the public signed 7-bit comb of the same template, whose LDS address is a digit, is refused by both analyses."""
    asm = assembly(tmp_path_factory, "k_gcomb")
    kern = "20k_gc_base_windowed_sILi5ELb1ELi256E"
    rep = ct_check.check(asm, kern, allow_global_loads=0, allow_lds_reads=True)
    assert rep["instructions"] > 2400 and rep["lds_reads"] == 64 and rep["scratch"] == 0 and rep["global_loads"] == []
    assert len(rep["branches"]) == 1 and re.match(r"s_cmpk?_(lg|eq|lt|gt|le|ge)_[iu]32 s\d+, \S+ ; s_cbranch_scc[01] ", rep["branches"][0]), rep["branches"]
    assert ct_check.check_after_secret_load(asm, kern) > 4000
    flow = ct_check.check_secret_flow(asm, kern, secret_args=[2, 4, 5, 6])             # (gcurve, order, k, table, ox, oy, oz, n)
    assert flow["secret_loads"] >= 1 and not flow["secret_lds"] and not flow["secret_scratch"]
    with pytest.raises(ct_check.Violation, match="LDS address"):
        ct_check.check(asm, "20k_gc_base_windowed_sILi7ELb0ELi1024E", allow_global_loads=0, allow_lds_reads=True)
    with pytest.raises(ct_check.Violation, match="LDS address"):
        ct_check.check_secret_flow(asm, "20k_gc_base_windowed_sILi7ELb0ELi1024E", secret_args=[2])


def test_registered_curve_constant_time_variable_base_window_loop(tmp_path_factory):
    """k_gvw_mult<true> (scalar_mult(ALG_WINDOWED | ALG_CONSTANT_TIME) on a registered curve's variable base, k_gvarwin.hip): the built-in constant-time
    loop's checks -- per window 32 global loads (the lane's eight entries, a 128-byte line at a time) at loop-invariant addresses, one branch on the window
    counter, nothing branching on a lane mask after the scalar is loaded -- and the secret-flow analysis with the scalar and the Jacobian result secret: no
    secret reaches an address, a branch, EXEC at a memory access or a lane-crossing instruction.  The default loop, whose one entry per window is addressed
    by a digit, is refused by both."""
    asm = assembly(tmp_path_factory, "k_gvarwin")
    kern = "10k_gvw_multILb1E"
    rep = ct_check.check(asm, kern, allow_global_loads=32)
    assert rep["instructions"] > 7000 and len(rep["global_loads"]) == 32 and rep["scratch"] == 0
    assert all(g.startswith("global_load_dwordx4 ") for g in rep["global_loads"])
    assert len(rep["branches"]) == 1 and re.match(r"s_cmpk?_(lg|eq|lt|gt|le|ge)_[iu]32 s\d+, \S+ ; s_cbranch_scc[01] ", rep["branches"][0]), rep["branches"]
    assert ct_check.check_after_secret_load(asm, kern) > 7000
    flow = ct_check.check_secret_flow(asm, kern, secret_args=[2, 6, 7, 8])             # (gcurve, order, k, k_stride, table, zg, ox, oy, oz, n): the scalar, the Jacobian k P
    assert flow["secret_loads"] >= 1 and not flow["secret_lds"] and not flow["secret_scratch"]
    with pytest.raises(ct_check.Violation, match="address"):
        ct_check.check(asm, "10k_gvw_multILb0E", allow_global_loads=32)
    with pytest.raises(ct_check.Violation, match="address"):
        ct_check.check_secret_flow(asm, "10k_gvw_multILb0E", secret_args=[2])
