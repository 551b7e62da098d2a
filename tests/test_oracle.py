"""CPU suite, part 1: pin the oracle (oracle/ecsimd_oracle.c) before anything trusts it.

(1) against every known-answer vector the reference's own tests hold (tests/golden/reference_kats.json),
(2) against lane-distinct vectors minted from the real reference (tests/golden/ref_vectors.json),
(3) against the live reference library where oracle/_ref exists,
(4) against an independent big-int model.
"""
import numpy as np
import pytest

from helpers import (CURVE_PARAMS, CURVE_NAMES, R, P256, SECP256K1, hexes_to_arr, arr_to_hexes, from_int, to_int,
                     ints_to_arr, arr_to_ints, from_hex, ec_mul, jacobian_mgry_to_affine_int, fill_random_np, splitmix64, SEED)

CURVES = [P256, SECP256K1]


def H(h, words=4):
    return hexes_to_arr([h], words)


def z128(h):          # a 128-bit literal of tests/ops.cpp zero-extended to 256 bits
    return H(h.rjust(64, "0"))


# ---------------------------------------------------------------- (1) the reference's own KATs
def test_kats_ops128(oracle, kats):
    k = kats["ops128"]
    for c in k["add"]:
        s, _ = oracle.add(z128(c["a"]), z128(c["b"]))
        assert to_int(s[0]) % 2**128 == int(c["r"], 16), c["src"]
    for c in k["sub"]:
        s, _ = oracle.sub(z128(c["a"]), z128(c["b"]))
        assert to_int(s[0]) % 2**128 == int(c["r"], 16), c["src"]
    for c in k["sub_if_above"]:
        assert to_int(oracle.sub_if_above(z128(c["a"]), z128(c["p"]))[0]) == int(c["r"], 16), c["src"]
    for c in k["mul"]:
        assert to_int(oracle.mul(z128(c["a"]), z128(c["b"]))[0]) == int(c["r"], 16), c["src"]
    for c in k["limb_mul"]:
        assert to_int(oracle.mul(z128(c["a"]), ints_to_arr([c["b"]]))[0]) == int(c["r"], 16), c["src"]
    for c in k["square"]:
        assert to_int(oracle.square(z128(c["a"]))[0]) == int(c["r"], 16), c["src"]
    for c in k["lt"]:
        _, bw = oracle.sub(z128(c["a"]), z128(c["b"]))
        assert int(bw[0]) == c["r"], c["src"]
    for c in k["shift_left_one"]:
        # 128-bit vectors: shift the value placed in the TOP half so the carry-out is the 128-bit one
        a = int(c["a"], 16) << 128
        s, cy = oracle.shift_left_one(ints_to_arr([a]))
        assert int(cy[0]) == c["carry"], c["src"]
        assert (to_int(s[0]) >> 128) == int(c["r"], 16), c["src"]


def test_kats_ops256(oracle, kats):
    k = kats["ops256"]
    for c in k["mul"]:
        assert to_int(oracle.mul(H(c["a"]), H(c["b"]))[0]) == int(c["r"], 16), c["src"]
    for c in k["mod_add_secp256k1"]:
        assert to_int(oracle.mod_add(SECP256K1, H(c["a"]), H(c["b"]))[0]) == int(c["r"], 16), c["src"]
    for c in k["mod_sub_secp256k1"]:
        assert to_int(oracle.mod_sub(SECP256K1, H(c["a"]), H(c["b"]))[0]) == int(c["r"], 16), c["src"]
    for c in k["mod_shift_left_one_secp256k1"]:
        assert to_int(oracle.mod_shift_left(SECP256K1, H(c["a"]), 1)[0]) == int(c["r"], 16), c["src"]


def test_kats_mgry(oracle, kats):
    k = kats["mgry_secp256k1"]; cv = SECP256K1; p = int(kats["secp256k1_p"], 16)
    vals = hexes_to_arr(k["from_to_roundtrip"]["values"])
    assert np.array_equal(oracle.mgry_to_classical(cv, oracle.mgry_from_classical(cv, vals)), vals)
    for a, b in k["reduce_equals_montgomery_mul"]["pairs"]:
        got = to_int(oracle.mgry_reduce(cv, oracle.mul(H(a), H(b)))[0])
        assert got == int(a, 16) * int(b, 16) * pow(R, -1, p) % p
    o = k["ops"]
    ma, mb = oracle.mgry_from_classical(cv, H(o["a"])), oracle.mgry_from_classical(cv, H(o["b"]))
    cl = lambda m: format(to_int(oracle.mgry_to_classical(cv, m)[0]), "064x")
    assert cl(oracle.mod_add(cv, ma, mb)) == o["a_plus_b"]
    assert cl(oracle.mod_sub(cv, ma, mb)) == o["a_minus_b"]
    assert cl(oracle.mod_sub(cv, mb, ma)) == o["b_minus_a"]
    for c in o["pow"]:
        assert cl(oracle.mgry_pow(cv, ma, from_hex(c["e"]))) == c["r"], c["e"]
    g = k["gfp"]
    assert cl(oracle.gfp_inverse(cv, oracle.mgry_from_classical(cv, H(g["inverse"]["a"])))) == g["inverse"]["r"]
    s, ok = oracle.gfp_sqrt(cv, oracle.mgry_from_classical(cv, H(g["sqrt"]["a"])))
    assert ok[0] == 1 and cl(s) == g["sqrt"]["r"]
    m = oracle.mgry_from_classical(cv, H(g["opposite_sum_zero"]["a"]))
    assert to_int(oracle.mod_add(cv, m, oracle.gfp_opposite(cv, m))[0]) == 0


def test_kats_p256_group(oracle, kats):
    k = kats["p256"]; cv = P256; c = oracle.constants(cv)
    y, ok = oracle.compute_y(cv, H(k["from_x"]["x"]))
    assert ok[0] == 1 and format(to_int(y[0]), "064x") == k["from_x"]["y"]
    G = oracle.from_affine(cv, c["gx"][None], c["gy"][None])
    # affine -> Jacobian -> affine round trip (tests/curve_point.cpp:28-41)
    ax, ay = oracle.to_affine(cv, G)
    assert np.array_equal(ax[0], c["gx"]) and np.array_equal(ay[0], c["gy"])
    aff = lambda J: tuple(format(to_int(v[0]), "064x") for v in oracle.to_affine(cv, J))
    dbl, Gu = oracle.dblu(cv, G)
    assert np.array_equal(dbl[2], Gu[2])                               # co-Z invariant (curve_group.cpp:45)
    assert aff(dbl) == (k["2G"]["x"], k["2G"]["y"]) and aff(Gu) == aff(G)
    tri, Gu2 = oracle.zaddu(cv, Gu, dbl)
    assert np.array_equal(tri[2], Gu2[2]) and aff(tri) == (k["3G"]["x"], k["3G"]["y"])
    tri2, _ = oracle.trplu(cv, G)
    assert aff(tri2) == (k["3G"]["x"], k["3G"]["y"])
    five, Gu3 = oracle.zdau(cv, dbl, Gu)                               # 2*(2G) + G (curve_group.cpp:85)
    assert np.array_equal(five[2], Gu3[2]) and aff(five) == (k["5G"]["x"], k["5G"]["y"])
    for s in k["scalar_mult"]:
        J = oracle.scalar_mult(cv, H(s["k"]), c["gx"][None], c["gy"][None])
        assert aff(J) == (s["x"], s["y"]), s["src"]


def test_survey_constants(oracle, kats):
    for name, cv in (("p256", P256), ("secp256k1", SECP256K1)):
        s = kats["survey_8c_constants"][name]; c = oracle.constants(cv)
        for key in ("r_p", "rsq_p", "pm1_r_p", "am", "bm"):
            assert format(to_int(c[key]), "064x") == s[key], (name, key)
        assert format(c["mprime"], "08x") == s["mprime"]
        G = oracle.from_affine(cv, c["gx"][None], c["gy"][None])
        assert format(to_int(G[0][0]), "064x") == s["gx_m"] and format(to_int(G[1][0]), "064x") == s["gy_m"]
        p = CURVE_PARAMS[cv]["p"]
        assert to_int(c["p_m2"]) == p - 2 and to_int(c["p_sqrt"]) == (p + 1) // 4
    s = kats["survey_8c_constants"]["p256"]; c = oracle.constants(P256)
    J = oracle.scalar_mult(P256, H(s["jacobian_k"]), c["gx"][None], c["gy"][None])
    assert [format(to_int(v[0]), "064x") for v in J] == [s["jacobian_X"], s["jacobian_Y"], s["jacobian_Z"]]
    s = kats["survey_8c_constants"]["secp256k1"]; c = oracle.constants(SECP256K1)
    for key, kk in (("5G", 5), ("k_0bc1", 0x0bc1b1f28709decb543d9677d2cc9942348f6b984deff409430740942ff38827),
                    ("k_0a89", 0x0a891cecc2bf13b0aca744434a9c9f4bd7bf5c8ed86e2f76e7df72bad813bd80)):
        ax, ay = oracle.to_affine(SECP256K1, oracle.scalar_mult(SECP256K1, ints_to_arr([kk]), c["gx"][None], c["gy"][None]))
        assert [format(to_int(ax[0]), "064x"), format(to_int(ay[0]), "064x")] == s[key]


# ---------------------------------------------------------------- (2) reference-minted fixtures
def run_against_golden(impl, golden, cvs=CURVES):
    """Shared by the oracle test (here) and the GPU test: `impl` has the oracle.loader method names."""
    b = golden["bignum"]; a_, b_ = hexes_to_arr(b["a"]), hexes_to_arr(b["b"])
    s, c = impl.add(a_, b_); assert arr_to_hexes(s) == b["add"] and c.tolist() == b["add_carry"]
    s, c = impl.sub(a_, b_); assert arr_to_hexes(s) == b["sub"] and c.tolist() == b["sub_borrow"]
    assert arr_to_hexes(impl.sub_if_above(a_, b_)) == b["sub_if_above"]
    s, c = impl.shift_left_one(a_); assert arr_to_hexes(s) == b["shift_left_one"] and c.tolist() == b["shift_carry"]
    assert arr_to_hexes(impl.mul(a_, b_), 8) == b["mul"] and arr_to_hexes(impl.square(a_), 8) == b["square"]
    for cv in cvs:
        g = golden["curves"][CURVE_NAMES[cv]]; f = g["field"]
        a, bb, t8 = hexes_to_arr(f["a"]), hexes_to_arr(f["b"]), hexes_to_arr(f["t8"], 8)
        assert arr_to_hexes(impl.mod_add(cv, a, bb)) == f["mod_add"]
        assert arr_to_hexes(impl.mod_sub(cv, a, bb)) == f["mod_sub"]
        for cnt in (1, 2, 3):
            assert arr_to_hexes(impl.mod_shift_left(cv, a, cnt)) == f["shl%d" % cnt]
        assert arr_to_hexes(impl.mgry_mul(cv, a, bb)) == f["mgry_mul"]
        assert arr_to_hexes(impl.mgry_sqr(cv, a)) == f["mgry_sqr"]
        assert arr_to_hexes(impl.mgry_reduce(cv, t8)) == f["mgry_reduce"]
        assert arr_to_hexes(impl.mgry_from_classical(cv, a)) == f["from_classical"]
        assert arr_to_hexes(impl.mgry_to_classical(cv, a)) == f["to_classical"]
        assert arr_to_hexes(impl.gfp_inverse(cv, a)) == f["inverse"]
        assert arr_to_hexes(impl.gfp_opposite(cv, a)) == f["opposite"]
        assert arr_to_hexes(impl.mgry_pow(cv, a, from_hex(f["pow_exponent"]))) == f["pow"]
        s, ok = impl.gfp_sqrt(cv, hexes_to_arr(f["mgry_sqr"]))
        assert all(f["sqrt_ok"]) and ok.tolist() == f["sqrt_ok"] and arr_to_hexes(s) == f["sqrt_of_sqr"]
        sm = g["scalar_mult_G"]; k = hexes_to_arr(sm["k"]); n = len(k)
        cst = g["constants"]; gx = np.tile(from_hex(cst["gx"]), (n, 1)); gy = np.tile(from_hex(cst["gy"]), (n, 1))
        J = impl.scalar_mult(cv, k, gx, gy)
        assert [arr_to_hexes(v) for v in J] == [sm["X"], sm["Y"], sm["Z"]]
        ax, ay = impl.to_affine(cv, J)
        assert arr_to_hexes(ax) == sm["ax"] and arr_to_hexes(ay) == sm["ay"]
        sv = g["scalar_mult_var"]; bx, by = hexes_to_arr(sv["bx"]), hexes_to_arr(sv["by"])
        J = impl.scalar_mult(cv, hexes_to_arr(sv["k"]), bx, by)
        assert [arr_to_hexes(v) for v in J] == [sv["X"], sv["Y"], sv["Z"]]
        P = impl.from_affine(cv, bx, by); fa = g["from_affine"]
        assert [arr_to_hexes(v) for v in P] == [fa["X"], fa["Y"], fa["Z"]]
        Rr, Pu = impl.dblu(cv, P)
        assert [arr_to_hexes(v) for v in Rr] == g["dblu"]["r"] and [arr_to_hexes(v) for v in Pu] == g["dblu"]["p_updated"]
        R3, Pu2 = impl.zaddu(cv, Pu, Rr)
        assert [arr_to_hexes(v) for v in R3] == g["zaddu"]["r"] and [arr_to_hexes(v) for v in Pu2] == g["zaddu"]["p_updated"]
        Rt, Put = impl.trplu(cv, P)
        assert [arr_to_hexes(v) for v in Rt] == g["trplu"]["r"] and [arr_to_hexes(v) for v in Put] == g["trplu"]["p_updated"]
        Rz, Qu = impl.zdau(cv, Rt, Put)
        assert [arr_to_hexes(v) for v in Rz] == g["zdau"]["r"] and [arr_to_hexes(v) for v in Qu] == g["zdau"]["q_updated"]
        Ra = impl.add_z2_1(cv, Rz, (P[0], P[1]))
        assert [arr_to_hexes(v) for v in Ra] == g["add_z2_1"]["r"]
        tx, ty = impl.to_affine(cv, Ra)
        assert arr_to_hexes(tx) == g["to_affine"]["x"] and arr_to_hexes(ty) == g["to_affine"]["y"]
        if "compute_y" in g:
            y, ok = impl.compute_y(cv, hexes_to_arr(g["compute_y"]["x"]))
            assert ok.tolist() == g["compute_y"]["ok"] and arr_to_hexes(y) == g["compute_y"]["y"]


def run_against_curve_fixture(impl, g, cv):
    """One curve block of tests/golden/ref_curves_vectors.json (the reference's curve_group<Curve> instantiated for a curve that is not its own) against
    `impl` under ITS id `cv` of that curve.  Shared by the oracle test below and the GPU test (tests/test_gpu_curves.py)."""
    f = g["field"]; a, bb = hexes_to_arr(f["a"]), hexes_to_arr(f["b"])
    for name in ("mod_add", "mod_sub", "mgry_mul"):
        assert arr_to_hexes(getattr(impl, name)(cv, a, bb)) == f[name], name
    for name, key in (("mgry_sqr", "mgry_sqr"), ("mgry_from_classical", "from_classical"), ("mgry_to_classical", "to_classical"), ("gfp_inverse", "inverse"), ("gfp_opposite", "opposite")):
        assert arr_to_hexes(getattr(impl, name)(cv, a)) == f[key], name
    sm = g["scalar_mult_G"]; k = hexes_to_arr(sm["k"]); n = len(k)
    cst = g["constants"]; gx = np.tile(from_hex(cst["gx"]), (n, 1)); gy = np.tile(from_hex(cst["gy"]), (n, 1))
    J = impl.scalar_mult(cv, k, gx, gy)
    assert [arr_to_hexes(v) for v in J] == [sm["X"], sm["Y"], sm["Z"]]
    ax, ay = impl.to_affine(cv, J)
    assert arr_to_hexes(ax) == sm["ax"] and arr_to_hexes(ay) == sm["ay"]
    sv = g["scalar_mult_var"]; bx, by = hexes_to_arr(sv["bx"]), hexes_to_arr(sv["by"])
    J = impl.scalar_mult(cv, hexes_to_arr(sv["k"]), bx, by)
    assert [arr_to_hexes(v) for v in J] == [sv["X"], sv["Y"], sv["Z"]]
    ax, ay = impl.to_affine(cv, J)
    assert arr_to_hexes(ax) == sv["ax"] and arr_to_hexes(ay) == sv["ay"]
    P = impl.from_affine(cv, bx, by); fa = g["from_affine"]
    assert [arr_to_hexes(v) for v in P] == [fa["X"], fa["Y"], fa["Z"]]
    Rr, Pu = impl.dblu(cv, P)
    assert [arr_to_hexes(v) for v in Rr] == g["dblu"]["r"] and [arr_to_hexes(v) for v in Pu] == g["dblu"]["p_updated"]
    R3, Pu2 = impl.zaddu(cv, Pu, Rr)
    assert [arr_to_hexes(v) for v in R3] == g["zaddu"]["r"] and [arr_to_hexes(v) for v in Pu2] == g["zaddu"]["p_updated"]
    Rt, Put = impl.trplu(cv, P)
    assert [arr_to_hexes(v) for v in Rt] == g["trplu"]["r"] and [arr_to_hexes(v) for v in Put] == g["trplu"]["p_updated"]
    Rz, Qu = impl.zdau(cv, Rt, Put)
    assert [arr_to_hexes(v) for v in Rz] == g["zdau"]["r"] and [arr_to_hexes(v) for v in Qu] == g["zdau"]["q_updated"]
    Ra = impl.add_z2_1(cv, Rz, (P[0], P[1]))
    assert [arr_to_hexes(v) for v in Ra] == g["add_z2_1"]["r"]
    tx, ty = impl.to_affine(cv, Ra)
    assert arr_to_hexes(tx) == g["to_affine"]["x"] and arr_to_hexes(ty) == g["to_affine"]["y"]
    if "compute_y" in g:
        y, ok = impl.compute_y(cv, hexes_to_arr(g["compute_y"]["x"]))
        assert ok.tolist() == g["compute_y"]["ok"] and arr_to_hexes(y) == g["compute_y"]["y"]


def curve_params_of(g):
    return {k: int(g["params"][k], 16) for k in ("p", "a", "b", "gx", "gy", "n")}


def test_oracle_on_registered_curves_matches_the_reference_fixtures(oracle, oracle_faithful, golden_curves):
    """Round 5: the oracle with a curve registered at run time (oracle_register_curve) against fixtures minted from the reference's curve_group<Curve>
    instantiated for that curve -- brainpoolP256r1 (dense p, a != -3), SM2 (a sparse p that is not P-256's), FRP256v1 (dense p, a = -3)."""
    for name, g in golden_curves["curves"].items():
        c = curve_params_of(g)
        for impl in (oracle, oracle_faithful):
            cv = impl.register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"])
            assert cv >= 2 and impl.register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"]) == cv
            cst = impl.constants(cv)
            for key, want in g["constants"].items():
                got = format(cst[key], "08x") if key == "mprime" else format(to_int(cst[key]), "064x")
                assert got == want, (name, key)
            run_against_curve_fixture(impl, g, cv)


def test_oracle_on_registered_curves_vs_live_reference(oracle_faithful, reference, golden_curves):
    """... and bug for bug against the compiled reference itself on random lane-distinct inputs (the square() defect included)."""
    from oracle.loader import REF_CURVES
    for name, c in REF_CURVES.items():
        cv, rv = oracle_faithful.register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"]), c["ref_id"]
        assert reference.register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"]) == rv
        rng = np.random.default_rng(500 + rv); n = 64
        rnd = lambda: rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
        gx, gy = np.tile(from_int(c["gx"]), (n, 1)), np.tile(from_int(c["gy"]), (n, 1))
        k = rnd()
        Jo, Jr = oracle_faithful.scalar_mult(cv, k, gx, gy, threads=4), reference.scalar_mult(rv, k, gx, gy, threads=4)
        for u, v in zip(Jo, Jr): assert np.array_equal(u, v), name
        bx, by = oracle_faithful.to_affine(cv, Jo)
        for u, v in zip((bx, by), reference.to_affine(rv, Jr)): assert np.array_equal(u, v)
        k2 = rnd()
        for u, v in zip(oracle_faithful.scalar_mult(cv, k2, bx, by, threads=4), reference.scalar_mult(rv, k2, bx, by, threads=4)): assert np.array_equal(u, v)
        P = oracle_faithful.from_affine(cv, bx, by)
        (Ro, Po), (Rr, Pr) = oracle_faithful.trplu(cv, P), reference.trplu(rv, P)
        for u, v in zip(Ro + Po, Rr + Pr): assert np.array_equal(u, v)
        (Zo, Qo), (Zr, Qr) = oracle_faithful.zdau(cv, Ro, Po), reference.zdau(rv, Rr, Pr)
        for u, v in zip(Zo + Qo, Zr + Qr): assert np.array_equal(u, v)


def test_oracle_matches_reference_fixtures(oracle, oracle_faithful, golden):
    run_against_golden(oracle, golden)            # the regular fixtures are defect-free (make_golden.py asserts it):
    run_against_golden(oracle_faithful, golden)   # both modes must reproduce them


def test_reference_square_defect_is_pinned(oracle, oracle_faithful, golden):
    """The reference's square() drops a carry on some inputs (mul.h:186-190, "TODO: carry?" at :207).
    FAITHFUL mode reproduces the reference bit for bit, wrong values included; EXACT mode returns a*a."""
    d = golden["square_defect"]; a = hexes_to_arr(d["a"])
    assert arr_to_hexes(oracle_faithful.square(a), 8) == d["reference_square"]
    assert arr_to_hexes(oracle.square(a), 8) == d["exact_square"]
    assert [format(v * v, "0128x") for v in arr_to_ints(a)] == d["exact_square"]
    wrong = [x != y for x, y in zip(d["reference_square"], d["exact_square"])]
    assert wrong == [bool(v) for v in d["reference_is_wrong"]] and sum(wrong) >= 8 and not all(wrong)
    assert arr_to_hexes(oracle_faithful.mgry_sqr(P256, a)) == d["p256_reference_mgry_sqr"]
    assert arr_to_hexes(oracle.mgry_sqr(P256, a)) == d["p256_exact_mgry_sqr"] == arr_to_hexes(oracle.mgry_mul(P256, a, a))
    oracle.reset_dropped_carries(); oracle.square(a)
    assert oracle.dropped_carries() == sum(wrong)                  # the event counter sees exactly the wrong ones


def test_oracle_scalar_mult_1s_fixture(oracle, golden):
    # scalar_mult_1s (curve_group.h:221-251) must equal scalar_mult with the scalar splat to all lanes
    for cv in CURVES:
        g = golden["curves"][CURVE_NAMES[cv]]["scalar_mult_1s"]
        bx, by = hexes_to_arr(g["bx"]), hexes_to_arr(g["by"])
        k = np.tile(from_hex(g["k1"]), (len(bx), 1))
        J = oracle.scalar_mult(cv, k, bx, by)
        assert [arr_to_hexes(v) for v in J] == [g["X"], g["Y"], g["Z"]]


def check_moduli_fixtures(impl, register, data):
    """tests/golden/ref_moduli_vectors.json (the reference's field layer instantiated for curve-less moduli) against `impl`."""
    for name, g in data["moduli"].items():
        p = int(g["p"], 16); fid = register(p); f = g["field"]
        a, b, t8 = hexes_to_arr(f["a"]), hexes_to_arr(f["b"]), hexes_to_arr(f["t8"], 8)
        eq = lambda got, key: np.array_equal(got, hexes_to_arr(f[key])) or pytest.fail(f"{name}: {key}")
        eq(impl.mod_add(fid, a, b), "mod_add"); eq(impl.mod_sub(fid, a, b), "mod_sub")
        eq(impl.mod_shift_left(fid, a, 1), "shl1"); eq(impl.mod_shift_left(fid, a, 3), "shl3")
        eq(impl.mgry_mul(fid, a, b), "mgry_mul"); eq(impl.mgry_sqr(fid, a), "mgry_sqr"); eq(impl.mgry_reduce(fid, t8), "mgry_reduce")
        eq(impl.mgry_from_classical(fid, a), "from_classical"); eq(impl.mgry_to_classical(fid, a), "to_classical")
        eq(impl.gfp_inverse(fid, a), "inverse"); eq(impl.mgry_pow(fid, a, from_hex(f["pow_exponent"])), "pow")
        if "opposite" in f:
            eq(impl.gfp_opposite(fid, a), "opposite")
            s, ok = impl.gfp_sqrt(fid, hexes_to_arr(f["mgry_sqr"]))
            wide_ok = np.array(f["sqrt_ok"], dtype=bool)                       # the reference reports all-or-nothing per wide of 4
            assert np.array_equal(ok.astype(bool).reshape(-1, 4).all(axis=1).repeat(4), wide_ok), name
            assert np.array_equal(s[wide_ok], hexes_to_arr(f["sqrt_of_sqr"])[wide_ok]), name


def test_oracle_matches_the_moduli_fixtures(oracle, golden_moduli):
    """The restatement is as generic in the modulus as the reference (mgry_mul.h:84-121 mgry_reduce<P>, mgry_csts.h:15-35): both group orders,
    2^255 - 19, 2^256 - 1 (composite) and P-192's prime, on vectors minted from the compiled reference."""
    check_moduli_fixtures(oracle, oracle.register_modulus, golden_moduli)
    for name, g in golden_moduli["moduli"].items():
        c = oracle.constants(oracle.register_modulus(int(g["p"], 16)))
        for k, v in g["constants"].items():
            assert (format(c[k], "08x") if k == "mprime" else format(to_int(c[k]), "064x")) == v, (name, k)


def test_oracle_moduli_vs_live_reference(oracle, oracle_faithful, reference):
    """Random and structured operands, every field op, every modulus the reference is compiled for -- and the square defect with it."""
    from oracle.loader import REF_MODULI
    rng = np.random.default_rng(21)
    pat = np.array([0, 0xffffffff, 0x80000000, 0x7fffffff, 1, 0xfffffffe], dtype=np.uint64)
    for name, p in REF_MODULI.items():
        fo, fr = oracle.register_modulus(p), reference.register_modulus(p)
        n = 2000
        a = ints_to_arr([to_int(x) % p for x in rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)])
        b = ints_to_arr([to_int(x) % p for x in rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)])
        w = pat[rng.integers(0, len(pat), size=(n // 2, 8))]
        a[: n // 2] = ints_to_arr([v % p for v in arr_to_ints((w[:, 0::2] | (w[:, 1::2] << np.uint64(32))).astype(np.uint64))])
        for f in ("mod_add", "mod_sub", "mgry_mul"):
            assert np.array_equal(getattr(oracle, f)(fo, a, b), getattr(reference, f)(fr, a, b)), (name, f)
        for f in ("mgry_from_classical", "mgry_to_classical"):
            assert np.array_equal(getattr(oracle, f)(fo, a), getattr(reference, f)(fr, a)), (name, f)
        assert np.array_equal(oracle_faithful.mgry_sqr(fo, a), reference.mgry_sqr(fr, a)), name          # bug for bug
        assert np.array_equal(oracle_faithful.gfp_inverse(fo, a[:64]), reference.gfp_inverse(fr, a[:64])), name
        t8 = ints_to_arr([x * y for x, y in zip(arr_to_ints(a), arr_to_ints(rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)))], 8)
        assert np.array_equal(oracle.mgry_reduce(fo, t8), reference.mgry_reduce(fr, t8)), name
        # and the arithmetic itself, against big ints
        assert arr_to_ints(oracle.mgry_mul(fo, a, b)) == [x * y * pow(R, -1, p) % p for x, y in zip(arr_to_ints(a), arr_to_ints(b))]


# ---------------------------------------------------------------- (3) live reference, random inputs
def structured_words(n, seed=1):
    """Carry-heavy operands: each 32-bit digit is 0 / ff..f / 80..0 / 7f..f / 1 / ff..e, or random (1 in 4)."""
    rng = np.random.default_rng(seed)
    pat = np.array([0, 0xffffffff, 0x80000000, 0x7fffffff, 1, 0xfffffffe], dtype=np.uint64)
    w = pat[rng.integers(0, len(pat), size=(n, 8))]
    w = np.where(rng.integers(0, 4, size=(n, 8)) == 0, rng.integers(0, 2**32, size=(n, 8), dtype=np.uint64), w)
    return (w[:, 0::2] | (w[:, 1::2] << np.uint64(32))).astype(np.uint64)


def test_structured_operands_vs_live_reference(oracle, oracle_faithful, reference):
    a = structured_words(50000); b = np.roll(a, 7, axis=0)
    assert np.array_equal(oracle_faithful.square(a), reference.square(a))            # bug-for-bug
    assert np.array_equal(oracle.square(a), reference.mul(a, a))                     # exact == the reference's own mul
    assert np.array_equal(oracle.mul(a, b), reference.mul(a, b))
    for cv in CURVES:
        p = CURVE_PARAMS[cv]["p"]
        ar = ints_to_arr([v % p for v in arr_to_ints(a[:4000])]); br = ints_to_arr([v % p for v in arr_to_ints(b[:4000])])
        assert np.array_equal(oracle_faithful.mgry_sqr(cv, ar), reference.mgry_sqr(cv, ar))
        for nm in ("mgry_mul", "mod_add", "mod_sub"):
            assert np.array_equal(getattr(oracle, nm)(cv, ar, br), getattr(reference, nm)(cv, ar, br)), nm


def digit_pattern_operands():
    """Every 256-bit operand whose eight 32-bit digits are drawn from {0, 1, 7fffffff, 80000000, fffffffe, ffffffff}: 6^8 operands."""
    pat = np.array([0, 1, 0x7fffffff, 0x80000000, 0xfffffffe, 0xffffffff], dtype=np.uint64)
    w = pat[np.indices((6,) * 8).reshape(8, -1).T]
    return (w[:, 0::2] | (w[:, 1::2] << np.uint64(32))).astype(np.uint64)


def test_every_digit_pattern_vs_live_reference(oracle, oracle_faithful, reference):
    """1 679 616 operands, 602 082 of them squared wrongly by the reference (mul.h:186-190): the bug-for-bug restatement equals the
    compiled reference on every one, the exact mode equals the reference's own mul(a, a)."""
    a = digit_pattern_operands()
    f = oracle_faithful.square(a)
    assert np.array_equal(f, reference.square(a))
    e = oracle.square(a)
    assert np.array_equal(e, reference.mul(a, a))
    assert int((f != e).any(axis=1).sum()) == 602082


@pytest.mark.parametrize("cv", CURVES)
def test_oracle_vs_live_reference(oracle_faithful, reference, cv):
    oracle = oracle_faithful          # bug-for-bug mode: must equal the compiled reference on EVERY input
    rng = np.random.default_rng(100 + cv); n = 96
    p = CURVE_PARAMS[cv]["p"]
    rnd = lambda w=4: rng.integers(0, 2**64, size=(n, w), dtype=np.uint64)
    a = ints_to_arr([to_int(x) % p for x in rnd()]); b = ints_to_arr([to_int(x) % p for x in rnd()])
    for name in ("mod_add", "mod_sub", "mgry_mul"):
        assert np.array_equal(getattr(oracle, name)(cv, a, b), getattr(reference, name)(cv, a, b)), name
    for name in ("mgry_sqr", "mgry_from_classical", "mgry_to_classical", "gfp_inverse", "gfp_opposite"):
        assert np.array_equal(getattr(oracle, name)(cv, a), getattr(reference, name)(cv, a)), name
    t8 = rnd(8); t8[:, 7] >>= np.uint64(8)
    assert np.array_equal(oracle.mgry_reduce(cv, t8), reference.mgry_reduce(cv, t8))
    c = oracle.constants(cv); k = rnd()
    gx, gy = np.tile(c["gx"], (n, 1)), np.tile(c["gy"], (n, 1))
    Jo, Jr = oracle.scalar_mult(cv, k, gx, gy, threads=4), reference.scalar_mult(cv, k, gx, gy, threads=4)
    for u, v in zip(Jo, Jr): assert np.array_equal(u, v)
    bx, by = oracle.to_affine(cv, Jo)
    k2 = rnd()
    for u, v in zip(oracle.scalar_mult(cv, k2, bx, by, threads=4), reference.scalar_mult(cv, k2, bx, by, threads=4)): assert np.array_equal(u, v)
    # Montgomery-form entry (what scalar_mult_p256 receives)
    P = oracle.from_affine(cv, bx, by)
    for u, v in zip(oracle.scalar_mult(cv, k2, P[0], P[1], threads=4, mgry_in=True), reference.scalar_mult(cv, k2, bx, by, threads=4)): assert np.array_equal(u, v)
    k1 = rnd()[0]
    for u, v in zip(oracle.scalar_mult(cv, np.tile(k1, (n, 1)), bx, by, threads=4), reference.scalar_mult_1s(cv, k1, bx, by)): assert np.array_equal(u, v)


# ---------------------------------------------------------------- (4) independent big-int model
@pytest.mark.parametrize("cv", CURVES)
def test_oracle_vs_bigint_model(oracle, cv):
    c = CURVE_PARAMS[cv]; p = c["p"]; rng = np.random.default_rng(7 + cv); n = 12
    a = [int(x) % p for x in arr_to_ints(rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64))]
    b = [int(x) % p for x in arr_to_ints(rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64))]
    Rinv = pow(R, -1, p)
    assert arr_to_ints(oracle.mgry_mul(cv, ints_to_arr(a), ints_to_arr(b))) == [x * y * Rinv % p for x, y in zip(a, b)]
    assert arr_to_ints(oracle.mod_sub(cv, ints_to_arr(a), ints_to_arr(b))) == [(x - y) % p for x, y in zip(a, b)]
    ks = [5, c["n"] - 2, c["n"] + 3, 2**256 - 1]   # (k = n-1, n, 0 are degenerate in the co-Z ladder: level J only)
    ks = ks + arr_to_ints(rng.integers(0, 2**64, size=(4, 4), dtype=np.uint64))
    G = (c["gx"], c["gy"])
    J = oracle.scalar_mult(cv, ints_to_arr(ks), ints_to_arr([c["gx"]] * len(ks)), ints_to_arr([c["gy"]] * len(ks)))
    for i, k in enumerate(ks):
        got = jacobian_mgry_to_affine_int(cv, *(to_int(v[i]) for v in J))
        assert got == ec_mul(cv, k % c["n"], G), hex(k)


def test_fill_random_twin():
    z = fill_random_np(3, SEED, 1, first_index=5)
    assert int(z[2, 3]) == splitmix64(SEED ^ (1 << 56) ^ ((5 + 2) * 4 + 3))
    assert int(fill_random_np(2, SEED, 2, clear_top_bits=1)[:, 3].max()) < 2**63


def test_config0_ops_bench_batch8(oracle, oracle_faithful, reference):
    """BASELINE configs[0]: the workload of benchs/ops.cpp (add_256, mul_256, sqr_256, mgry_sqr_256,
    mgry_reduce_512 on the secp256k1 prime, :22-24,106-116) on batch = 8 = two wides, CPU only: the real
    reference against the restatement.  bench_mgry_sqr / bench_mgry_reduce zero the last BYTE (LastZero, :26-34)."""
    n = 8; rng = np.random.default_rng(8)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); b = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    t8 = rng.integers(0, 2**64, size=(n, 8), dtype=np.uint64)
    a0 = a.copy(); a0[:, 0] &= np.uint64(0xFFFFFFFFFFFFFF00); t8[:, 0] &= np.uint64(0xFFFFFFFFFFFFFF00)   # the last big-endian byte is limb 0's low byte
    s_o, c_o = oracle.add(a, b); s_r, c_r = reference.add(a, b)
    assert np.array_equal(s_o, s_r) and np.array_equal(c_o, c_r)
    assert np.array_equal(oracle.mul(a, b), reference.mul(a, b))
    assert np.array_equal(oracle_faithful.square(a), reference.square(a)) and np.array_equal(oracle.square(a), reference.mul(a, a))
    assert np.array_equal(oracle_faithful.mgry_sqr(SECP256K1, a0), reference.mgry_sqr(SECP256K1, a0))
    assert np.array_equal(oracle.mgry_reduce(SECP256K1, t8), reference.mgry_reduce(SECP256K1, t8))
