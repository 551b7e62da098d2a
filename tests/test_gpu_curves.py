"""GPU parity: curves registered at RUN time (round 5) -- the reference's curve_group<Curve> for any Curve (curve.h:12-15; curve_group.h:25-33, 64-87,
189-218), here ecsimd_hip_register_curve + the generic kernels (k_gcurve.hip, k_gladder.hip; the ladder's loop on fe29.cuh's 29-bit limbs with the
dense prime in SGPRs).  Level J like the built-in curves: X, Y, Z bit for bit against
  * fixtures minted from the REAL reference instantiated for brainpoolP256r1 / SM2 / FRP256v1 (tests/golden/ref_curves_vectors.json),
  * the oracle with the same curve registered (random, edge-scalar, digit-pattern inputs; ragged sizes),
  * the compiled reference itself (oracle/_ref: its curve ids 10, 11, 12) at 2^15 lanes, ECSIMD_HIP_REF_SQUARE_COMPAT included,
and -- the built-in curves' parameters registered with ECSIMD_HIP_CURVE_GENERIC_KERNELS -- against the special-form kernels, entry point by entry point.
"""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import CURVE_PARAMS, P256, SECP256K1, SEED, from_int, to_int, ints_to_arr
from oracle.loader import REF_CURVES
from test_oracle import run_against_curve_fixture, curve_params_of, digit_pattern_operands

pytestmark = pytest.mark.gpu
THREADS = min(16, os.cpu_count() or 1)
LADDER_RADIX32, REF_SQUARE_COMPAT, OUT_AFFINE, BASE_MGRY = 256, 64, 2, 1


@pytest.fixture(scope="module")
def gpu(engine):
    from gpu_adapter import EngineNP
    return EngineNP(engine)


def register(c, generic=False):
    from ecsimd_amd.engine import register_curve
    return register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c.get("n"), generic_kernels=generic)


def ids(oracle, c, generic=False):
    return register(c, generic), oracle.register_curve(c["p"], c["a"] % c["p"], c["b"], c["gx"], c["gy"])


def same(a, b):
    return all(np.array_equal(u, v) for u, v in zip(a, b))


def test_registered_curves_vs_the_reference_fixtures(gpu, engine, golden_curves):
    """Every block of ref_curves_vectors.json through the C ABI on a registered curve id: field ops under the curve id as a field id, from_affine /
    to_affine / compute_y, DBLU ZADDU TRPLU ZDAU ADD_Z2_1, the ladder on 32 edge and random scalars (Jacobian AND affine), scalar_mult_1s."""
    for name, g in golden_curves["curves"].items():
        c = curve_params_of(g)
        cid = register(c)
        assert cid >= 0x10000 and register(c) == cid
        run_against_curve_fixture(gpu, g, cid)
        # the same ladder inputs on canonical words (LADDER_RADIX32) and with the reference's squaring (the fixtures are clear of its defect)
        sv = g["scalar_mult_var"]
        k, bx, by = (engine.to_device(ints_to_arr([int(h, 16) for h in sv[key]])) for key in ("k", "bx", "by"))
        want = [ints_to_arr([int(h, 16) for h in sv[key]]) for key in ("X", "Y", "Z")]
        for fl in (LADDER_RADIX32, REF_SQUARE_COMPAT):
            assert same([engine.to_numpy(t) for t in engine.scalar_mult(cid, k, bx, by, flags=fl)], want), (name, fl)
        ax, ay = engine.scalar_mult(cid, k, bx, by, flags=OUT_AFFINE)
        assert [format(to_int(v), "064x") for v in engine.to_numpy(ax)] == sv["ax"] and [format(to_int(v), "064x") for v in engine.to_numpy(ay)] == sv["ay"]
        xo, none = engine.scalar_mult(cid, k, bx, by, flags=OUT_AFFINE, x_only=True)
        assert none is None and np.array_equal(engine.to_numpy(xo), engine.to_numpy(ax))
        s1 = g["scalar_mult_1s"]
        J = engine.scalar_mult_1s(cid, from_int(int(s1["k1"], 16)), engine.to_device(ints_to_arr([int(h, 16) for h in s1["bx"]])), engine.to_device(ints_to_arr([int(h, 16) for h in s1["by"]])))
        assert [[format(to_int(v), "064x") for v in engine.to_numpy(t)] for t in J] == [s1["X"], s1["Y"], s1["Z"]]
        sm = g["scalar_mult_G"]
        J = engine.scalar_mult_base(cid, engine.to_device(ints_to_arr([int(h, 16) for h in sm["k"]])))
        assert [[format(to_int(v), "064x") for v in engine.to_numpy(t)] for t in J] == [sm["X"], sm["Y"], sm["Z"]]


@pytest.mark.parametrize("name", list(REF_CURVES))
def test_registered_curve_ladder_vs_oracle_on_edge_and_random_scalars(gpu, engine, oracle, name):
    """The ladder on a registered curve, all three loops (29-bit limbs, canonical words, reference squaring), against the oracle with the same curve
    registered: every edge scalar of the built-in curves' test (0, 1, n - 1, n, n + 1, 2^256 - n +- 1, 2^256 - 1, 2^255 ...: level J, degenerate
    outputs included), random 256-bit scalars, lane-distinct base points, ragged batch sizes (1, 3, 257), Montgomery-form and classical base points."""
    c = REF_CURVES[name]
    cid, oid = ids(oracle, c)
    order = c["n"]
    edge = [0, 1, 2, 3, order - 2, order - 1, order, order + 1, 2**256 - order - 2, 2**256 - order - 1, 2**256 - order, 2**256 - order + 1, 2**256 - 1, 2**255, 2**255 - 1,
            (order - 1) // 2, (order + 1) // 2, 4, 6, 2**200]
    n = 1024 + 257
    rng = np.random.default_rng(sum(name.encode()))
    k = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    k[:len(edge)] = ints_to_arr(edge)
    s = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    gx, gy = np.tile(from_int(c["gx"]), (n, 1)), np.tile(from_int(c["gy"]), (n, 1))
    bx, by = oracle.to_affine(oid, oracle.scalar_mult(oid, s, gx, gy, threads=THREADS))            # lane-distinct points
    gbx, gby = gpu.scalar_mult(cid, s, gx, gy, affine=True)
    assert np.array_equal(gbx, bx) and np.array_equal(gby, by)
    exp = oracle.scalar_mult(oid, k, bx, by, threads=THREADS)
    for fl in (0, LADDER_RADIX32, REF_SQUARE_COMPAT):
        got = [engine.to_numpy(t) for t in engine.scalar_mult(cid, engine.to_device(k), engine.to_device(bx), engine.to_device(by), flags=fl)]
        bad = np.flatnonzero((got[0] != exp[0]).any(axis=1) | (got[1] != exp[1]).any(axis=1) | (got[2] != exp[2]).any(axis=1))
        assert bad.size == 0, (name, fl, bad[:8])
    P = oracle.from_affine(oid, bx, by)
    assert same(gpu.scalar_mult(cid, k, P[0], P[1], mgry_in=True), exp)
    for m in (1, 3, 257):
        assert same(gpu.scalar_mult(cid, k[:m], bx[:m], by[:m]), [v[:m] for v in exp])
    assert same(gpu.scalar_mult_base(cid, k[:300]), oracle.scalar_mult(oid, k[:300], gx[:300], gy[:300], threads=THREADS))
    # affine output through the shared inversion = the oracle's per-lane to_affine where Z != 0; (0, 0) where Z = 0
    ax, ay = oracle.to_affine(oid, exp)
    gax, gay = gpu.scalar_mult(cid, k, bx, by, affine=True)
    assert np.array_equal(gax, ax) and np.array_equal(gay, ay)
    empty = engine.scalar_mult(cid, engine.empty(0), engine.empty(0), engine.empty(0))
    assert all(t.shape[0] == 0 for t in empty)


@pytest.mark.parametrize("name", list(REF_CURVES))
def test_registered_curve_formulas_on_digit_pattern_coordinates(gpu, oracle, name):
    """DBLU / ZADDU / ZDAU / ADD_Z2_1 / TRPLU as expression DAGs over GF(p) on the carry-heavy digit-pattern family (level J does not need curve points),
    compute_y (generalised to a != -3 exactly as the oracle is), on_curve, and ecsimd_hip_zdau_repeat in both radices against the oracle's ZDAU iterated."""
    c = REF_CURVES[name]
    cid, oid = ids(oracle, c)
    a = digit_pattern_operands()[::97]
    pp = np.tile(from_int(c["p"]), (len(a), 1))
    red = lambda v: oracle.sub_if_above(oracle.sub_if_above(v, pp), pp)
    x, y = red(a), red(np.roll(a, 1234, axis=0))
    P = oracle.from_affine(oid, x, y)
    assert same(gpu.from_affine(cid, x, y), P)
    (R, Pu), (Rg, Pug) = oracle.dblu(oid, P), gpu.dblu(cid, P)
    assert same(Rg + Pug, R + Pu), "DBLU"
    (R3, Pu2), (R3g, Pu2g) = oracle.zaddu(oid, Pu, R), gpu.zaddu(cid, Pu, R)
    assert same(R3g + Pu2g, R3 + Pu2), "ZADDU"
    (Rt, Put), (Rtg, Putg) = oracle.trplu(oid, P), gpu.trplu(cid, P)
    assert same(Rtg + Putg, Rt + Put), "TRPLU"
    (Rz, Qu), (Rzg, Qug) = oracle.zdau(oid, R3, Pu2), gpu.zdau(cid, R3, Pu2)
    assert same(Rzg + Qug, Rz + Qu), "ZDAU"
    assert same(gpu.add_z2_1(cid, Rz, (P[0], P[1])), oracle.add_z2_1(oid, Rz, (P[0], P[1]))), "ADD_Z2_1"
    assert same(gpu.to_affine(cid, Rz), oracle.to_affine(oid, Rz))
    yo, oko = oracle.compute_y(oid, x)
    yg, okg = gpu.compute_y(cid, x)
    assert np.array_equal(okg, oko) and np.array_equal(yg[oko == 1], yo[oko == 1]) and 0 < int(oko.sum()) < len(x)
    on = gpu.e.to_numpy(gpu.e.on_curve(cid, gpu._up(x[oko == 1]), gpu._up(yo[oko == 1])))
    assert on.all() and not gpu.e.to_numpy(gpu.e.on_curve(cid, gpu._up(x[oko == 1]), gpu._up(x[oko == 1]))).all()
    m = 512
    P0, Q0 = tuple(v[:m] for v in Rt), tuple(v[:m] for v in Put)
    for iters, swap in ((1, 0), (1, 1), (2, 0b10), (7, 0b1011001), (67, 0xdeadbeefcafef00d)):
        Pc, Qc = P0, Q0
        for t in range(iters):
            Rr, Qn = oracle.zdau(oid, Pc, Qc)
            Pc, Qc = (Qn, Rr) if (swap >> (t & 63)) & 1 else (Rr, Qn)
        exp = (Pc[0], Pc[1], Qc[0], Qc[1], Pc[2])
        for radix in (29, 32):
            assert same(gpu.zdau_repeat(cid, P0, (Q0[0], Q0[1]), iters, swap, radix), exp), (radix, iters, hex(swap))


@pytest.mark.parametrize("name", list(REF_CURVES))
def test_registered_curve_vs_the_live_reference(engine, reference, name):
    """2^15 lanes against the compiled reference instantiated for this curve (oracle/_ref, its ids 10 / 11 / 12): with ECSIMD_HIP_REF_SQUARE_COMPAT not one
    lane may differ; the default loop (exact squaring) may differ only where the reference's square() dropped a carry (~3e-6 of lanes)."""
    c = REF_CURVES[name]
    cid, rid = register(c), c["ref_id"]
    n = 1 << 15
    k = engine.fill_random(n, SEED, 40 + rid); s = engine.fill_random(n, SEED, 41 + rid)
    bx, by = engine.scalar_mult_base(cid, s, flags=OUT_AFFINE)
    kn, xn, yn = (engine.to_numpy(t) for t in (k, bx, by))
    ref = reference.scalar_mult(rid, kn, xn, yn, threads=THREADS)
    got = [engine.to_numpy(t) for t in engine.scalar_mult(cid, k, bx, by, flags=REF_SQUARE_COMPAT)]
    assert same(got, ref), name
    got = [engine.to_numpy(t) for t in engine.scalar_mult(cid, k, bx, by)]
    differing = int(np.count_nonzero((got[0] != ref[0]).any(axis=1) | (got[1] != ref[1]).any(axis=1) | (got[2] != ref[2]).any(axis=1)))
    assert differing <= 3, (name, differing)
    engine.lib.ecsimd_hip_set_ref_square_compat(engine.ctx, C.c_int(1))               # the context option: every entry point of the DAG
    try:
        P = engine.from_affine(cid, bx, by)
        R = engine.trplu(cid, P)
        want = reference.trplu(rid, reference.from_affine(rid, xn, yn))
        assert same([engine.to_numpy(t) for t in R], want[0]) and same([engine.to_numpy(t) for t in P], want[1])
        ax, ay = engine.scalar_mult(cid, k[:4096].contiguous(), bx[:4096].contiguous(), by[:4096].contiguous(), flags=OUT_AFFINE)
        rx, ry = reference.to_affine(rid, [v[:4096] for v in ref])
        assert np.array_equal(engine.to_numpy(ax), rx) and np.array_equal(engine.to_numpy(ay), ry)
    finally:
        engine.lib.ecsimd_hip_set_ref_square_compat(engine.ctx, C.c_int(0))


@pytest.mark.parametrize("cv", [P256, SECP256K1])
def test_builtin_curves_through_the_generic_kernels_equal_the_special_form_kernels(gpu, engine, cv):
    """P-256 and secp256k1 registered like any other curve (ECSIMD_HIP_CURVE_GENERIC_KERNELS) must return, entry point by entry point and
    bit for bit, what their special-form kernels return: 2^16 lane-distinct points and scalars, every formula, both ladders' loops, both output forms."""
    c = dict(CURVE_PARAMS[cv]); c["a"] %= c["p"]
    gid = register(c, generic=True)
    assert gid >= 0x10000 and register(c) == cv                          # without the flag the built-in id
    n = 1 << 16
    k = engine.fill_random(n, SEED, 60 + cv); s = engine.fill_random(n, SEED, 61 + cv)
    bx, by = engine.scalar_mult_base(cv, s, flags=OUT_AFFINE)
    gbx, gby = engine.scalar_mult_base(gid, s, flags=OUT_AFFINE)
    assert np.array_equal(engine.to_numpy(gbx), engine.to_numpy(bx)) and np.array_equal(engine.to_numpy(gby), engine.to_numpy(by))
    for fl in (0, LADDER_RADIX32, OUT_AFFINE):
        assert same([engine.to_numpy(t) for t in engine.scalar_mult(gid, k, bx, by, flags=fl)], [engine.to_numpy(t) for t in engine.scalar_mult(cv, k, bx, by, flags=fl)]), fl
    assert same([engine.to_numpy(t) for t in engine.scalar_mult(gid, k, bx, by, flags=REF_SQUARE_COMPAT)], [engine.to_numpy(t) for t in engine.scalar_mult(cv, k, bx, by, flags=REF_SQUARE_COMPAT)])
    k1 = engine.to_numpy(k)[7]
    assert same([engine.to_numpy(t) for t in engine.scalar_mult_1s(gid, k1, bx, by)], [engine.to_numpy(t) for t in engine.scalar_mult_1s(cv, k1, bx, by)])
    xn, yn = engine.to_numpy(bx), engine.to_numpy(by)
    P = gpu.from_affine(cv, xn, yn)
    assert same(gpu.from_affine(gid, xn, yn), P)
    (R, Pu), (Rg, Pug) = gpu.dblu(cv, P), gpu.dblu(gid, P)
    assert same(Rg + Pug, R + Pu)
    (R3, Pu2), (R3g, Pu2g) = gpu.zaddu(cv, Pu, R), gpu.zaddu(gid, Pu, R)
    assert same(R3g + Pu2g, R3 + Pu2)
    (Rz, Qu), (Rzg, Qug) = gpu.zdau(cv, R3, Pu2), gpu.zdau(gid, R3, Pu2)
    assert same(Rzg + Qug, Rz + Qu)
    assert same(gpu.add_z2_1(gid, Rz, (P[0], P[1])), gpu.add_z2_1(cv, Rz, (P[0], P[1])))
    assert same(gpu.trplu(gid, P)[0], gpu.trplu(cv, P)[0])
    assert same(gpu.to_affine(gid, Rz), gpu.to_affine(cv, Rz))
    yc, okc = gpu.compute_y(cv, xn); yg, okg = gpu.compute_y(gid, xn)
    assert okc.all() and np.array_equal(okg, okc) and np.array_equal(yg, yc)
    for radix in (29, 32):
        assert same(gpu.zdau_repeat(gid, R3, (Pu2[0], Pu2[1]), 9, 0b101100111, radix), gpu.zdau_repeat(cv, R3, (Pu2[0], Pu2[1]), 9, 0b101100111, radix))


def test_registered_curve_entry_points_that_do_not_exist_say_so(engine, oracle):
    """The table-driven ALG_* algorithms exist for the two built-in curves: a registered curve id is refused there (BAD_ARG with a message), never silently
    served by some other curve's kernels; ECDSA on a registered curve needs the group order it was registered with."""
    c = REF_CURVES["brainpoolP256r1"]
    cid = register(c)
    from ecsimd_amd import ALG_WINDOWED, ALG_CONSTANT_TIME, EcsimdHipError
    k = engine.fill_random(8, SEED, 1)
    bx, by = engine.scalar_mult_base(cid, k, flags=OUT_AFFINE)
    from ecsimd_amd import ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG
    from ecsimd_amd import ALG_NO_ENDOMORPHISM
    for fl in (OUT_AFFINE | ALG_WINDOWED_SIGNED, OUT_AFFINE | ALG_WINDOWED | ALG_NO_ENDOMORPHISM, OUT_AFFINE | ALG_CONSTANT_TIME):
        with pytest.raises(EcsimdHipError, match="a registered curve has"):
            engine.scalar_mult(cid, k, bx, by, flags=fl)                  # a variable base: the window loop (k_gvarwin.hip; plain and constant-time) and the ladder, nothing else
    with pytest.raises(EcsimdHipError, match="OUT_AFFINE"):
        engine.scalar_mult(cid, k, bx, by, flags=ALG_WINDOWED)            # the window loop's Jacobian representative is not the reference's
    for fl in (OUT_AFFINE | ALG_CONSTANT_TIME, OUT_AFFINE | ALG_WINDOWED_SIGNED | ALG_CONSTANT_TIME, OUT_AFFINE | ALG_WINDOWED_SIGNED | ALG_WINDOWED, OUT_AFFINE | ALG_WINDOWED_BIG | ALG_CONSTANT_TIME):
        with pytest.raises(EcsimdHipError, match="a registered curve has"):
            engine.scalar_mult_base(cid, k, flags=fl)
    with pytest.raises(EcsimdHipError, match="OUT_AFFINE"):
        engine.scalar_mult_base(cid, k, flags=ALG_WINDOWED)                # the comb's Jacobian representative is not the reference's
    with pytest.raises(EcsimdHipError):
        engine.scalar_mult_base(0x10000 + 4000, k)                       # no such curve
    # ECDSA needs the group order: the same curve registered WITHOUT n (a different record is impossible -- the parameters are the key -- so: another curve)
    from ecsimd_amd.engine import register_curve
    c2 = REF_CURVES["sm2"]
    # (registered with n elsewhere in this process or not, a curve id without an order must refuse; a fresh curve: y^2 = x^3 + a x + b' through another point)
    p_ = c2["p"]; gx2 = 5; rhs = None
    for gx2 in range(5, 200):
        rhs = (gx2 ** 3 + c2["a"] * gx2 + c2["b"]) % p_
        gy2 = pow(rhs, (p_ + 1) // 4, p_)
        if gy2 * gy2 % p_ == rhs:
            break
    noorder = register_curve(p_, c2["a"], c2["b"], gx2, gy2)              # SM2's curve with another base point and no order
    with pytest.raises(EcsimdHipError, match="group order"):
        engine.ecdsa_verify(noorder, k, k, k, bx, by)
    with pytest.raises(EcsimdHipError, match="group order"):
        engine.ecdsa_sign(noorder, k, k, k)
    with pytest.raises(EcsimdHipError, match="group order"):
        engine.scalar_mult_base(noorder, k, flags=OUT_AFFINE | ALG_WINDOWED)   # the comb recodes modulo n
    with pytest.raises(EcsimdHipError, match="group order"):
        engine.scalar_mult(noorder, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)   # and so does the window loop
    # ... and the window loop of a VARIABLE base needs that order to be a prime in p's Hasse interval (then every valid point has order n): brainpoolP256r1 with
    # n + 2 = 3 * ... keeps the comb of its generator (a statement about multiples of G alone) and the ladder, and refuses the per-lane tables
    from ecsimd_amd.engine import curve_capabilities, CURVE_COMB, CURVE_WINDOW_VARIABLE_BASE
    composite = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"] + 2)
    assert curve_capabilities(composite) & (CURVE_COMB | CURVE_WINDOW_VARIABLE_BASE) == CURVE_COMB and curve_capabilities(cid) & CURVE_WINDOW_VARIABLE_BASE
    with pytest.raises(EcsimdHipError, match="prime order"):
        engine.scalar_mult(composite, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    lx, ly = engine.scalar_mult(composite, k, bx, by, flags=OUT_AFFINE)                    # the ladder has no such condition
    wx, wy = engine.scalar_mult(cid, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    assert np.array_equal(engine.to_numpy(lx), engine.to_numpy(wx)) and np.array_equal(engine.to_numpy(ly), engine.to_numpy(wy))


# ---------------------------------------------------------------- the first application on a registered curve: ECDSA, u1 G + u2 Q, SEC1 (round 5)
def _affine_model(c):
    """Textbook affine arithmetic on Python integers for the curve c (None = infinity): the third opinion next to the oracle and the reference."""
    p, a = c["p"], c["a"]

    def add(P, Q):
        if P is None: return Q
        if Q is None: return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if (y1 + y2) % p == 0: return None
            lam = (3 * x1 * x1 + a) * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    def mul(k, P):
        R = None
        for bit in bin(k)[2:] if k else "":
            R = add(R, R)
            if bit == "1": R = add(R, P)
        return R
    return add, mul


@pytest.mark.parametrize("name", list(REF_CURVES))
def test_ecdsa_and_double_scalar_mult_on_a_registered_curve(engine, name):
    """ecsimd_hip_ecdsa_sign / _verify / double_scalar_mult / affine_add with a registered curve id: two passes of the reference's ladder (the scalars kept clear of
    its three degenerate values by k_gc_ladder_safe_scalars), one shared inversion each, a batched affine addition, the arithmetic modulo the curve's own n.
    Against textbook affine arithmetic on Python integers: signatures from (e, d, k) incl. the nonces at which the ladder alone is wrong (n - 1, 2^256 - n,
    2^256 - n - 1), u1 G + u2 Q incl. those scalars, Q = +-G (tangent, infinity); a 2^14 sign -> verify round trip; every input tampered in turn."""
    c = REF_CURVES[name]
    cid = register(c)
    p, n_, G = c["p"], c["n"], (c["gx"], c["gy"])
    add, mul = _affine_model(c)
    up = engine.to_device
    rng = np.random.default_rng(sum(name.encode()) + 5)
    rnd = lambda: int.from_bytes(rng.bytes(32), "big") % (n_ - 1) + 1
    special = [k for k in (n_ - 1, 2**256 - n_, 2**256 - n_ - 1, 1, 2, n_ - 2) if 0 < k < n_]
    m = 40
    d = [rnd() for _ in range(m)]; k = special + [rnd() for _ in range(m - len(special))]; e = [int.from_bytes(rng.bytes(32), "big") for _ in range(m)]
    r_, s_, ok = (engine.to_numpy(t) for t in engine.ecdsa_sign(cid, up(ints_to_arr(e)), up(ints_to_arr(d)), up(ints_to_arr(k))))
    Q = [mul(di, G) for di in d]
    for i in range(m):
        rr = mul(k[i], G)[0] % n_
        ss = pow(k[i], -1, n_) * (e[i] + rr * d[i]) % n_
        assert ok[i] == 1 and (to_int(r_[i]), to_int(s_[i])) == (rr, ss), (name, i, hex(k[i]))
    qx, qy = up(ints_to_arr([q[0] for q in Q])), up(ints_to_arr([q[1] for q in Q]))
    assert engine.to_numpy(engine.ecdsa_verify(cid, up(ints_to_arr(e)), up(r_), up(s_), qx, qy)).all()
    # u1 G + u2 Q on scalars the ladder alone gets wrong, and on sums that are tangents or infinity
    u1 = special + [5, 7, 0, 9] + [rnd() for _ in range(8)]
    u2 = [3] * len(special) + [n_ - 5, 7, 4, 0] + special[:2] + [rnd() for _ in range(6)]
    Qs = [Q[i] for i in range(len(special))] + [G, G, G, G] + [Q[20 + i] for i in range(8)]
    mm = len(u1)
    rx, ry, fin = (engine.to_numpy(t) for t in engine.double_scalar_mult(cid, up(ints_to_arr(u1)), up(ints_to_arr(u2[:mm])), up(ints_to_arr([q[0] for q in Qs])), up(ints_to_arr([q[1] for q in Qs]))))
    for i in range(mm):
        want = add(mul(u1[i], G), mul(u2[i], Qs[i]))
        assert (bool(fin[i]), (to_int(rx[i]), to_int(ry[i]))) == ((want is not None), want if want is not None else (0, 0)), (name, i)
    sx, sy, sf = (engine.to_numpy(t) for t in engine.affine_add(cid, (qx[:8].contiguous(), qy[:8].contiguous()), (qx[:8].contiguous(), qy[:8].contiguous())))
    assert all((to_int(sx[i]), to_int(sy[i])) == add(Q[i], Q[i]) and sf[i] == 1 for i in range(8))
    # a larger round trip, then every input tampered in turn
    N = 1 << 14
    tile = lambda v: np.tile(ints_to_arr(v), ((N + len(v) - 1) // len(v), 1))[:N].copy()
    dd = tile(d); kk = engine.to_numpy(engine.fill_random(N, SEED, 90)); kk[:, 3] >>= np.uint64(2); ee = engine.to_numpy(engine.fill_random(N, SEED, 91))
    R, S, OK = engine.ecdsa_sign(cid, up(ee), up(dd), up(kk))
    assert bool(OK.all())
    QX, QY = up(tile([q[0] for q in Q])), up(tile([q[1] for q in Q]))
    assert bool(engine.ecdsa_verify(cid, up(ee), R, S, QX, QY).all())
    bad_e = ee.copy(); bad_e[:, 0] ^= np.uint64(1)
    assert not engine.to_numpy(engine.ecdsa_verify(cid, up(bad_e), R, S, QX, QY)).any()
    Rn, Sn = engine.to_numpy(R).copy(), engine.to_numpy(S).copy()
    t = Rn.copy(); t[:, 0] ^= np.uint64(2)
    assert not engine.to_numpy(engine.ecdsa_verify(cid, up(ee), up(t), S, QX, QY)).any()
    t = Sn.copy(); t[:, 1] ^= np.uint64(4)
    assert not engine.to_numpy(engine.ecdsa_verify(cid, up(ee), R, up(t), QX, QY)).any()
    assert not engine.to_numpy(engine.ecdsa_verify(cid, up(ee), R, S, up(np.roll(engine.to_numpy(QX), 1, axis=0)), up(np.roll(engine.to_numpy(QY), 1, axis=0))))[:len(d)].all()
    zero = np.zeros_like(Rn); order = np.tile(from_int(n_), (N, 1))
    for rr, ss in ((zero, Sn), (Rn, zero), (order, Sn), (Rn, order)):
        assert not engine.to_numpy(engine.ecdsa_verify(cid, up(ee), up(rr), up(ss), QX, QY)).any()
    offx = engine.to_numpy(QX).copy(); offx[:, 0] ^= np.uint64(1)                                   # off the curve (or another point: either way no valid signature)
    assert not engine.to_numpy(engine.ecdsa_verify(cid, up(ee), R, S, up(offx), QY)).any()
    assert not engine.to_numpy(engine.ecdsa_verify(cid, up(ee), R, S, up(zero), up(zero))).any()     # Q = (0, 0): the point at infinity
    low_s = ints_to_arr([(n_ - to_int(v)) for v in Sn[:64]])                                        # (r, n - s) verifies too: ECDSA's malleability, as on the built-in curves
    assert engine.to_numpy(engine.ecdsa_verify(cid, up(ee[:64].copy()), up(Rn[:64].copy()), up(low_s), QX[:64].contiguous(), QY[:64].contiguous())).all()
    engine.ecdsa_sign(cid, up(ee[:4096].copy()), up(dd[:4096].copy()), up(kk[:4096].copy()))          # nothing nonce-derived stays behind a signing call
    assert not engine.workspace_bytes()[:9 * 4096 * 32 + 2 * 4096].any()                             # capi.hip gc_plan: adjusted nonce, Jacobian k G, affine x, the flags


@pytest.mark.parametrize("name", list(REF_CURVES))
def test_sec1_codecs_on_a_registered_curve(engine, oracle, name):
    """SEC 1 2.3.3 / 2.3.4 on a registered curve: uncompressed and compressed round trips of 4 099 lane-distinct points, the square-root branch with the
    curve's own a, per-lane validity (x >= p, x not on the curve, a wrong prefix, y off by one)."""
    c = REF_CURVES[name]
    cid, oid = ids(oracle, c)
    n = 4099
    s = engine.fill_random(n, SEED, 77)
    x, y = engine.scalar_mult_base(cid, s, flags=OUT_AFFINE)
    for compressed in (False, True):
        rec = engine.sec1_encode(cid, x, y, compressed)
        dx, dy, ok = engine.sec1_decode(cid, rec, compressed)
        assert bool(ok.all()) and np.array_equal(engine.to_numpy(dx), engine.to_numpy(x)) and np.array_equal(engine.to_numpy(dy), engine.to_numpy(y)), (name, compressed)
    rec = engine.sec1_encode(cid, x, y, True).cpu().numpy().copy()
    yo, oko = oracle.compute_y(oid, engine.to_numpy(x))
    assert oko.all()
    rec[0, 0] = 0x05                                                       # a prefix that is none
    rec[1, 1:] = np.frombuffer((c["p"] + 1).to_bytes(32, "big"), dtype=np.uint8) if c["p"] + 1 < 2**256 else rec[1, 1:]        # x >= p
    xs = engine.to_numpy(x)
    bump = 2
    while True:                                                             # an x with no point on the curve
        xv = (to_int(xs[2]) + bump) % c["p"]; rhs = (xv ** 3 + c["a"] * xv + c["b"]) % c["p"]
        if pow(rhs, (c["p"] - 1) // 2, c["p"]) != 1:
            break
        bump += 1
    rec[2, 1:] = np.frombuffer(xv.to_bytes(32, "big"), dtype=np.uint8)
    import torch
    _, _, ok = engine.sec1_decode(cid, torch.from_numpy(rec).to(engine.tdev), True)
    okn = engine.to_numpy(ok)
    assert not okn[:3].any() and okn[3:].all()
    full = engine.sec1_encode(cid, x, y, False).cpu().numpy().copy()
    full[5, 64] ^= 1                                                        # y off by one
    _, _, ok = engine.sec1_decode(cid, torch.from_numpy(full).to(engine.tdev), False)
    okn = engine.to_numpy(ok)
    assert okn[5] == 0 and okn[:5].all() and okn[6:].all()


@pytest.mark.parametrize("name", list(REF_CURVES))
def test_generator_comb_on_a_registered_curve(engine, name):
    """scalar_mult_base(ALG_WINDOWED [| ALG_CONSTANT_TIME] | OUT_AFFINE) and (ALG_WINDOWED_SIGNED | OUT_AFFINE) with a registered curve id (k_gcomb.hip: the
    4-bit and the signed 7-bit odd-digit table of multiples of the curve's generator in LDS, built from the reference's ladder on first use): the ladder's affine points lane for lane on 2^17 + 77 random 256-bit scalars,
    the true k G (textbook affine arithmetic on Python integers) on the edge scalars -- the comb's own exceptional scalar k* = n - 2 (n mod 16) and its images,
    0 and n (infinity: (0, 0)), and the three scalars at which the LADDER is wrong -- and the x-only form."""
    c = REF_CURVES[name]
    cid = register(c)
    n_ = c["n"]
    add, mul = _affine_model(c)
    G = (c["gx"], c["gy"])
    ks = n_ - 2 * (n_ % 16)
    ks7 = n_ - 2 * (n_ % 2**252)                                             # the signed 7-bit comb's (summed from the bottom)
    ks20 = n_ - 2 * (n_ % 2**240)                                            # the 20-bit comb's
    edge = [0, 1, 2, 3, 15, 16, 17, 31, 32, 33, n_ - 2, n_ - 1, n_, n_ + 1, n_ + 2, ks, ks - 1, ks + 1, n_ - ks, ks + n_ if ks + n_ < 2**256 else ks, 2**256 - n_ - 1, 2**256 - n_,
            ks7, ks7 - 1, ks7 + 1, ks7 - 2, ks7 + 2, n_ - ks7, n_ - ks7 + 1, 127, 128, 129, 2**252 - 1, 2**252 + 1, 15 * 2**252 + 1, 2**7 - 1, 2**14 + 1,
            ks20, ks20 - 1, ks20 + 1, n_ - ks20, 2**20 - 1, 2**20, 2**20 + 1, 2**240 - 1, 2**240 + 1, 65535 * 2**240 + 1, 2**21 - 1,
            2**256 - n_ + 1, 2**256 - 1, 2**255, 2**255 - 1, (n_ - 1) // 2, (n_ + 1) // 2, 0x1111111111111111111111111111111111111111111111111111111111111111, 2**252, 16**63, 15 * 16**63]
    N = (1 << 17) + 77
    rng = np.random.default_rng(sum(name.encode()) + 9)
    k = rng.integers(0, 2**64, size=(N, 4), dtype=np.uint64)
    k[:len(edge)] = ints_to_arr(edge)
    kd = engine.to_device(k)
    lx, ly = (engine.to_numpy(t) for t in engine.scalar_mult_base(cid, kd, flags=OUT_AFFINE))
    from ecsimd_amd import ALG_WINDOWED, ALG_CONSTANT_TIME, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG
    for fl in (ALG_WINDOWED, ALG_WINDOWED | ALG_CONSTANT_TIME, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG):   # (4-bit; constant-time 5-bit; signed 7-bit in 148 KiB of LDS; 20-bit in 436 MB of device memory)
        wx, wy = (engine.to_numpy(t) for t in engine.scalar_mult_base(cid, kd, flags=OUT_AFFINE | fl))
        for i, kv in enumerate(edge):
            want = mul(kv % n_, G)
            assert (to_int(wx[i]), to_int(wy[i])) == (want if want is not None else (0, 0)), (name, fl, i, hex(kv))
        assert np.array_equal(wx[len(edge):], lx[len(edge):]) and np.array_equal(wy[len(edge):], ly[len(edge):]), (name, fl)
        xo, none = engine.scalar_mult_base(cid, kd, flags=OUT_AFFINE | fl, x_only=True)
        assert none is None and np.array_equal(engine.to_numpy(xo), wx)
    for m in (1, 3, 255, 257, 1023, 1025):                                  # ragged batches: a partial workgroup still loads the whole table
        for fl in (ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG):
            wx, wy = (engine.to_numpy(t) for t in engine.scalar_mult_base(cid, engine.to_device(k[70:70 + m].copy()), flags=OUT_AFFINE | fl))
            assert np.array_equal(wx, lx[70:70 + m]) and np.array_equal(wy, ly[70:70 + m])
    assert all(t.shape[0] == 0 for t in engine.scalar_mult_base(cid, engine.empty(0), flags=OUT_AFFINE | ALG_WINDOWED))


@pytest.mark.parametrize("name", list(REF_CURVES))
def test_variable_base_window_loop_on_a_registered_curve(engine, name):
    """scalar_mult(ALG_WINDOWED | OUT_AFFINE) with a registered curve id (k_gvarwin.hip: the lane's eight odd multiples of P over one Z, the window loop in
    modified Jacobian coordinates on the isomorphic curve -- a general a, a dense prime -- and its ALG_CONSTANT_TIME form): the ladder's affine points lane for lane on 2^16 + 77 random 256-bit
    scalars and lane-distinct base points; the true k P (textbook affine arithmetic on Python integers) on the edge scalars -- 0 and n (infinity: (0, 0)), the
    three scalars at which the LADDER is wrong, digit patterns that make every window's digit +-1 / +-15; Montgomery-form base points, one shared scalar,
    the x-only form, ragged and empty batches, and an invalid base point that must not touch its neighbours."""
    c = REF_CURVES[name]
    cid = register(c)
    n_ = c["n"]
    add, mul = _affine_model(c)
    from ecsimd_amd import ALG_WINDOWED, BASE_MGRY
    edge = [0, 1, 2, 3, 15, 16, 17, 31, 32, 33, n_ - 2, n_ - 1, n_, n_ + 1, n_ + 2, 2**256 - n_ - 1, 2**256 - n_, 2**256 - n_ + 1, 2**256 - 1, 2**255, 2**255 - 1,
            (n_ - 1) // 2, (n_ + 1) // 2, 0x1111111111111111111111111111111111111111111111111111111111111111, 0xffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff0f % n_,
            0x0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f, 0x1010101010101010101010101010101010101010101010101010101010101011, 2**252, 16**63, 15 * 16**63]
    N = (1 << 16) + 77
    rng = np.random.default_rng(sum(name.encode()) + 21)
    k = rng.integers(0, 2**64, size=(N, 4), dtype=np.uint64)
    k[:len(edge)] = ints_to_arr(edge)
    kd = engine.to_device(k)
    s = engine.fill_random(N, SEED, 41)
    bx, by = engine.scalar_mult_base(cid, s, flags=OUT_AFFINE)                                         # lane-distinct base points
    lx, ly = (engine.to_numpy(t) for t in engine.scalar_mult(cid, kd, bx, by, flags=OUT_AFFINE))      # the reference's ladder
    wx, wy = (engine.to_numpy(t) for t in engine.scalar_mult(cid, kd, bx, by, flags=OUT_AFFINE | ALG_WINDOWED))
    bxn, byn = engine.to_numpy(bx), engine.to_numpy(by)
    for i, kv in enumerate(edge):
        want = mul(kv % n_, (to_int(bxn[i]), to_int(byn[i])))
        assert (to_int(wx[i]), to_int(wy[i])) == (want if want is not None else (0, 0)), (name, i, hex(kv))
    assert np.array_equal(wx[len(edge):], lx[len(edge):]) and np.array_equal(wy[len(edge):], ly[len(edge):]), name
    from ecsimd_amd import ALG_CONSTANT_TIME
    cx_, cy_ = (engine.to_numpy(t) for t in engine.scalar_mult(cid, kd, bx, by, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME))   # every entry read, lane masks: the same points
    assert np.array_equal(cx_, wx) and np.array_equal(cy_, wy), name
    c1x, c1y = (engine.to_numpy(t) for t in engine.scalar_mult_1s(cid, from_int(edge[-7]), bx[:777].contiguous(), by[:777].contiguous(), flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME))
    l1x, l1y = (engine.to_numpy(t) for t in engine.scalar_mult_1s(cid, from_int(edge[-7]), bx[:777].contiguous(), by[:777].contiguous(), flags=OUT_AFFINE))
    assert np.array_equal(c1x, l1x) and np.array_equal(c1y, l1y)
    for i in range(len(edge), len(edge) + 6):                                                            # and a third opinion on a few random lanes
        assert (to_int(wx[i]), to_int(wy[i])) == mul(to_int(k[i]) % n_, (to_int(bxn[i]), to_int(byn[i])))
    xo, none = engine.scalar_mult(cid, kd, bx, by, flags=OUT_AFFINE | ALG_WINDOWED, x_only=True)
    assert none is None and np.array_equal(engine.to_numpy(xo), wx)
    jx, jy, jz = engine.from_affine(cid, bx, by)                                                        # base points in Montgomery form
    mx, my = (engine.to_numpy(t) for t in engine.scalar_mult(cid, kd, jx, jy, flags=OUT_AFFINE | ALG_WINDOWED | BASE_MGRY))
    assert np.array_equal(mx, wx) and np.array_equal(my, wy)
    k1 = to_int(k[len(edge) + 3])
    sx, sy = (engine.to_numpy(t) for t in engine.scalar_mult_1s(cid, from_int(k1), bx[:4099].contiguous(), by[:4099].contiguous(), flags=OUT_AFFINE | ALG_WINDOWED))
    ox, oy = (engine.to_numpy(t) for t in engine.scalar_mult_1s(cid, from_int(k1), bx[:4099].contiguous(), by[:4099].contiguous(), flags=OUT_AFFINE))
    assert np.array_equal(sx, ox) and np.array_equal(sy, oy)
    for m in (1, 3, 255, 257):                                                                           # ragged batches
        a0 = 40
        rx, ry = (engine.to_numpy(t) for t in engine.scalar_mult(cid, engine.to_device(k[a0:a0 + m].copy()), bx[a0:a0 + m].contiguous(), by[a0:a0 + m].contiguous(), flags=OUT_AFFINE | ALG_WINDOWED))
        assert np.array_equal(rx, wx[a0:a0 + m]) and np.array_equal(ry, wy[a0:a0 + m])
    assert all(t.shape[0] == 0 for t in engine.scalar_mult(cid, engine.empty(0), engine.empty(0), engine.empty(0), flags=OUT_AFFINE | ALG_WINDOWED))
    # an invalid base point (0, 0) and one off the curve: their lanes are garbage or (0, 0), every other lane is what it was
    bad_x, bad_y = bxn[:600].copy(), byn[:600].copy()
    bad_x[77] = 0; bad_y[77] = 0
    bad_y[300, 0] ^= np.uint64(1)
    rx, ry = (engine.to_numpy(t) for t in engine.scalar_mult(cid, engine.to_device(k[:600].copy()), engine.to_device(bad_x), engine.to_device(bad_y), flags=OUT_AFFINE | ALG_WINDOWED))
    keep = np.ones(600, dtype=bool); keep[[77, 300]] = False
    assert np.array_equal(rx[keep], wx[:600][keep]) and np.array_equal(ry[keep], wy[:600][keep])
    assert not rx[77].any() and not ry[77].any()


def test_window_loops_of_a_registered_curve_at_the_full_batch_size(engine):
    """BASELINE configs[3]'s batch (2^24 scalar multiplications, on one GPU) through the registered curve's window loops, held by properties that need no
    oracle at that size: every lane equals the reference ladder's affine point; k P + (n - k) P is the point at infinity on every lane; a P + b P = (a + b) P
    through the batched affine addition; the constant-time loop returns the plain loop's points; u1 G + u2 Q (the signed comb + the window loop) equals the sum of
    its two halves.  brainpoolP256r1: a random coefficient a, a dense prime."""
    import torch
    from ecsimd_amd import ALG_WINDOWED, ALG_CONSTANT_TIME, ALG_WINDOWED_SIGNED
    c = REF_CURVES["brainpoolP256r1"]
    cid = register(c)
    N = 1 << 24
    nn = engine.to_device(np.tile(from_int(c["n"]), (N, 1)))
    s = engine.fill_random(N, SEED, 301)
    bx, by = engine.scalar_mult_base(cid, s, flags=OUT_AFFINE | ALG_WINDOWED_SIGNED)                     # lane-distinct base points
    k = engine.fill_random(N, SEED, 302, clear_top_bits=1)                                               # < 2^255 <= n
    w = engine.scalar_mult(cid, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    l = engine.scalar_mult(cid, k, bx, by, flags=OUT_AFFINE)
    assert torch.equal(w[0], l[0]) and torch.equal(w[1], l[1]), "the window loop and the ladder disagree somewhere in 2^24 lanes"
    del l
    ct = engine.scalar_mult(cid, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME)
    assert torch.equal(ct[0], w[0]) and torch.equal(ct[1], w[1])
    del ct
    nk, borrow = engine.sub(nn, k)
    assert not bool(borrow.any())
    m = engine.scalar_mult(cid, nk, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    zx, zy, fin = engine.affine_add(cid, w, m)
    assert not bool(fin.any()) and not bool(zx.any()) and not bool(zy.any()), "k P + (n - k) P is not the point at infinity somewhere"
    del m, nk, zx, zy, fin
    a = engine.fill_random(N, SEED, 303, clear_top_bits=2); b = engine.fill_random(N, SEED, 304, clear_top_bits=2)      # a + b < 2^255
    ab, carry = engine.add(a, b)
    assert not bool(carry.any())
    pa = engine.scalar_mult(cid, a, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    pb = engine.scalar_mult(cid, b, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    sx, sy, fin = engine.affine_add(cid, pa, pb)
    pab = engine.scalar_mult(cid, ab, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    assert bool(fin.all()) and torch.equal(sx, pab[0]) and torch.equal(sy, pab[1]), "a P + b P != (a + b) P somewhere"
    del pb, sx, sy, pab, ab
    ga = engine.scalar_mult_base(cid, b, flags=OUT_AFFINE | ALG_WINDOWED_SIGNED)                         # u1 G + u2 Q against its two halves
    ex, ey, efin = engine.affine_add(cid, ga, pa)
    dx, dy, dfin = engine.double_scalar_mult(cid, b, a, bx, by)
    assert torch.equal(dx, ex) and torch.equal(dy, ey) and torch.equal(dfin, efin)


@pytest.mark.parametrize("cv", [P256, SECP256K1])
def test_variable_base_window_loop_of_a_builtin_curve_through_the_generic_kernels(engine, cv):
    """P-256 (a = -3) / secp256k1 (a = 0) registered like any other curve: the generic window loop returns the built-in loops' affine points."""
    c = CURVE_PARAMS[cv]
    gid = register(c, generic=True)
    from ecsimd_amd import ALG_WINDOWED
    N = 1 << 14
    k = engine.fill_random(N, SEED, 124)
    kn = engine.to_numpy(k); kn[:4] = ints_to_arr([0, c["n"], c["n"] - 2, 2]); k = engine.to_device(kn)
    bx, by = engine.scalar_mult_base(cv, engine.fill_random(N, SEED, 125), flags=OUT_AFFINE)
    want = [engine.to_numpy(t) for t in engine.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)]
    from ecsimd_amd import ALG_CONSTANT_TIME
    for fl in (ALG_WINDOWED, ALG_WINDOWED | ALG_CONSTANT_TIME):
        assert same([engine.to_numpy(t) for t in engine.scalar_mult(gid, k, bx, by, flags=OUT_AFFINE | fl)], want), (cv, fl)


@pytest.mark.parametrize("cv", [P256, SECP256K1])
def test_generator_comb_of_a_builtin_curve_through_the_generic_kernels(engine, cv):
    """P-256 / secp256k1 registered like any other curve (ECSIMD_HIP_CURVE_GENERIC_KERNELS): the generic comb returns the built-in comb's affine points."""
    c = CURVE_PARAMS[cv]
    gid = register(c, generic=True)
    from ecsimd_amd import ALG_WINDOWED, ALG_CONSTANT_TIME
    k = engine.fill_random(1 << 16, SEED, 123)
    kn = engine.to_numpy(k); kn[:4] = ints_to_arr([0, c["n"], c["n"] - 2, 2]); k = engine.to_device(kn)       # P-256's k* = n - 2
    want = [engine.to_numpy(t) for t in engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | ALG_WINDOWED)]
    for fl in (ALG_WINDOWED, ALG_WINDOWED | ALG_CONSTANT_TIME):
        assert same([engine.to_numpy(t) for t in engine.scalar_mult_base(gid, k, flags=OUT_AFFINE | fl)], want), (cv, fl)


@pytest.mark.parametrize("name", list(REF_CURVES))
def test_small_base_batches_on_a_registered_curve_take_the_comb_and_keep_the_ladders_bits(engine, name):
    """scalar_mult_base(OUT_AFFINE) without an algorithm flag on up to 2^16 lanes of a registered curve goes through the constant-time comb of its generator
    (0.35 ms instead of a 1.8 ms ladder launch) and must return the LADDER's affine bits -- at the ladder's three degenerate scalars too (the lanes take the
    ladder's coordinates from the context's record).  Against the ladder itself run on G as a variable base, on every edge scalar, around the route's limit,
    x only, and through ecsimd_hip_scalar_mult with x = y = NULL | BASE_GENERATOR; an explicit ladder flag and the Jacobian form keep the ladder."""
    import torch
    from ecsimd_amd import BASE_GENERATOR
    c = REF_CURVES[name]
    cid = register(c)
    order = c["n"]
    edge = [0, 1, 2, 3, order - 2, order - 1, order, order + 1, 2**256 - order - 2, 2**256 - order - 1, 2**256 - order, 2**256 - order + 1,
            2**256 - 1, 2**255, 2**255 - 1, (order - 1) // 2, (order + 1) // 2, order - 2 * (order % 16), 31, 32, 2**250]
    for n in (1, 4, 100, 4096, 1 << 16, (1 << 16) + 1):
        k = engine.fill_random(n, SEED, 170 + c["ref_id"])
        m = min(n, len(edge))
        k[:m] = engine.to_device(ints_to_arr(edge[:m]))
        gx = engine.to_device(np.tile(from_int(c["gx"]), (n, 1))); gy = engine.to_device(np.tile(from_int(c["gy"]), (n, 1)))
        lx, ly = engine.scalar_mult(cid, k, gx, gy, flags=OUT_AFFINE)                 # the reference's ladder, G as a variable base
        bx, by = engine.scalar_mult_base(cid, k, flags=OUT_AFFINE)                    # the route under test (the ladder itself above 2^16)
        assert torch.equal(bx, lx) and torch.equal(by, ly), (name, n, np.flatnonzero((engine.to_numpy(bx) != engine.to_numpy(lx)).any(axis=1))[:8])
        fx, fy = engine.scalar_mult_base(cid, k, flags=OUT_AFFINE | LADDER_RADIX32)   # an explicit ladder flag keeps the ladder
        assert torch.equal(fx, lx) and torch.equal(fy, ly)
        xo, none = engine.scalar_mult_base(cid, k, flags=OUT_AFFINE, x_only=True)
        assert none is None and torch.equal(xo, lx), n
        if n <= 4096:
            ox, oy = engine.empty(n), engine.empty(n)
            engine._bind_stream()
            rc = engine.lib.ecsimd_hip_scalar_mult(engine.ctx, C.c_int(cid), C.c_void_p(k.data_ptr()), None, None, C.c_void_p(ox.data_ptr()), C.c_void_p(oy.data_ptr()), None,
                                                   C.c_size_t(n), C.c_int(OUT_AFFINE | BASE_GENERATOR))
            assert rc == 0 and torch.equal(ox, lx) and torch.equal(oy, ly)
    k = engine.fill_random(64, SEED, 172)
    gx = engine.to_device(np.tile(from_int(c["gx"]), (64, 1))); gy = engine.to_device(np.tile(from_int(c["gy"]), (64, 1)))
    assert all(torch.equal(a, b) for a, b in zip(engine.scalar_mult_base(cid, k), engine.scalar_mult(cid, k, gx, gy)))      # Jacobian: the ladder's representative


def test_device_group_on_a_registered_curve(engine):
    """ecsimd_hip_group_scalar_mult with a registered curve id (the registry is process-wide, every member's context sees the same id): three members on device 0,
    uneven shards, Jacobian and affine, gathered into member 0's arrays = one context's ladder on the whole batch; the host-array form too."""
    import torch
    from ecsimd_amd import DeviceGroup, shard_range_c
    c = REF_CURVES["brainpoolP256r1"]
    cid = register(c)
    n = 5003; G = 3
    grp = DeviceGroup([0] * G)
    try:
        k = engine.fill_random(n, SEED, 201); s_ = engine.fill_random(n, SEED, 202)
        bx, by = engine.scalar_mult_base(cid, s_, flags=OUT_AFFINE)
        exp = engine.scalar_mult(cid, k, bx, by)
        spans = [shard_range_c(n, m, G) for m in range(G)]
        cut = lambda t: [t[f:f + cnt].contiguous() for f, cnt in spans]
        got, _ = grp.scalar_mult(cid, cut(k), cut(bx), cut(by), n)
        assert all(torch.equal(a, b) for a, b in zip(got, exp))
        (ax, ay), _ = grp.scalar_mult(cid, cut(k), cut(bx), cut(by), n, flags=OUT_AFFINE)
        ex, ey = engine.to_affine(cid, exp)
        assert torch.equal(ax, ex) and torch.equal(ay, ey)
        H = grp.scalar_mult_host(cid, *(engine.to_numpy(t) for t in (k, bx, by)))
        assert all(np.array_equal(h, engine.to_numpy(e)) for h, e in zip(H, exp))
    finally:
        grp.close()


def test_registered_curve_entry_points_replay_from_a_hip_graph(engine):
    """The ladder, the variable-base window loop (plain and constant-time), the generator's comb (small-batch route and ALG_WINDOWED), u1 G + u2 Q and ecdsa_verify on a registered curve captured into one hipGraph after a
    warm-up call (which builds the comb's table and the record of the degenerate scalars and sizes the workspace); new inputs in place, replay = eager calls."""
    import torch
    from ecsimd_amd import ALG_WINDOWED
    c = REF_CURVES["frp256v1"]
    cid = register(c)
    n = 1 << 14
    k = engine.fill_random(n, SEED, 191, clear_top_bits=1); s = engine.fill_random(n, SEED, 192, clear_top_bits=1)
    bx, by = engine.scalar_mult_base(cid, s, flags=OUT_AFFINE)
    J = [engine.empty(n) for _ in range(3)]; A = [engine.empty(n) for _ in range(2)]; W = [engine.empty(n) for _ in range(2)]; VW = [engine.empty(n) for _ in range(2)]; VC = [engine.empty(n) for _ in range(2)]
    from ecsimd_amd import ALG_CONSTANT_TIME

    def run():
        engine.scalar_mult(cid, k, bx, by, out=J)
        engine.scalar_mult(cid, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED, out=VW + [None])                          # the lane's own window table: nothing to build beforehand
        engine.scalar_mult(cid, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME, out=VC + [None])
        engine.scalar_mult_base(cid, k, flags=OUT_AFFINE, out=A + [None])                       # <= 2^16 lanes: the constant-time comb + the patch
        engine.scalar_mult_base(cid, k, flags=OUT_AFFINE | ALG_WINDOWED, out=W + [None])
        return engine.double_scalar_mult(cid, s, k, bx, by), engine.ecdsa_verify(cid, s, k, s, bx, by)
    run(); torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            D, V = run()
    torch.cuda.synchronize()
    k.copy_(engine.fill_random(n, SEED, 193, clear_top_bits=1))
    g.replay(); torch.cuda.synchronize()
    got = [t.clone() for t in J + A + W + list(D) + [V]]
    evw = engine.scalar_mult(cid, k, bx, by, flags=OUT_AFFINE)                                      # the window loops' points = the ladder's affine points
    assert all(torch.equal(a, b) for a, b in zip(VW + VC, list(evw) + list(evw)))
    ej = engine.scalar_mult(cid, k, bx, by)
    ea = engine.scalar_mult_base(cid, k, flags=OUT_AFFINE | LADDER_RADIX32)                      # the ladder itself
    ed = engine.double_scalar_mult(cid, s, k, bx, by); ev = engine.ecdsa_verify(cid, s, k, s, bx, by)
    torch.cuda.synchronize()
    want = list(ej) + list(ea) + list(ea) + list(ed) + [ev]
    assert all(torch.equal(a, b) for a, b in zip(got, want))


def _is_prime(n_):
    if n_ < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n_ % q == 0:
            return n_ == q
    d, r = n_ - 1, 0
    while d % 2 == 0:
        d //= 2; r += 1
    for a_ in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):                    # deterministic below 3.3e24, a strong probable-prime test above
        x = pow(a_, d, n_)
        if x in (1, n_ - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n_
            if x == n_ - 1:
                break
        else:
            return False
    return True


def _random_curve(bits, seed, a_kind):
    """A curve nobody compiled anything for: a random prime p = 3 mod 4 of `bits` bits, random a (or -3, 0), random b with 4a^3 + 27b^2 != 0, a point on it."""
    import random
    rng = random.Random(seed)
    while True:
        p_ = rng.getrandbits(bits) | (1 << (bits - 1)) | 3
        if _is_prime(p_):
            break
    while True:
        a_ = {"random": rng.randrange(p_), "-3": p_ - 3, "0": 0}[a_kind]
        b_ = rng.randrange(1, p_)
        if (4 * a_ ** 3 + 27 * b_ * b_) % p_ == 0:
            continue
        for _ in range(200):
            x = rng.randrange(p_)
            rhs = (x ** 3 + a_ * x + b_) % p_
            y = pow(rhs, (p_ + 1) // 4, p_)
            if y * y % p_ == rhs and y != 0:
                return dict(p=p_, a=a_, b=b_, gx=x, gy=y)


@pytest.mark.parametrize("bits,a_kind", [(256, "random"), (256, "-3"), (256, "0"), (255, "random"), (254, "random"), (233, "random"), (224, "-3"), (192, "random"), (129, "random"),
                                         (128, "0"), (64, "random"), (61, "random"), (33, "random"), (32, "random"), (31, "-3"), (17, "random")])
def test_curves_nobody_compiled_anything_for(gpu, engine, oracle, bits, a_kind):
    """"Any curve" means any: random primes p = 3 mod 4 from 17 to 256 bits -- the top limbs of the 29-bit representation empty, p one bit short of 2^256, p with
    its top bit set -- with a random a, a = -3 and a = 0, registered WITHOUT a group order (the reference's concept has none), against the oracle with the same
    curve registered: the point formulas on points of the curve, the ladder's three loops on random and edge scalars (level J: X, Y, Z), the affine conversion
    through the shared inversion, on_curve / compute_y, and what needs an order is refused."""
    c = _random_curve(bits, 1000 * bits + len(a_kind), a_kind)
    cid = register(c)
    oid = oracle.register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"])
    n = 300
    rng = np.random.default_rng(bits)
    k = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    k[:12] = ints_to_arr([0, 1, 2, 3, 4, 5, 2**256 - 1, 2**255, c["p"], c["p"] - 1, c["p"] + 1 if c["p"] + 1 < 2**256 else 7, 2**200 + 1])
    gx, gy = np.tile(from_int(c["gx"]), (n, 1)), np.tile(from_int(c["gy"]), (n, 1))
    s = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    B = oracle.scalar_mult(oid, s, gx, gy, threads=THREADS)                                      # lane-distinct points s G (Jacobian; a few may be degenerate: Z = 0)
    bx, by = oracle.to_affine(oid, B)
    live = ~((bx == 0).all(axis=1) & (by == 0).all(axis=1))                                      # small groups: s G may be infinity
    bx, by, k, gx, gy, s = (v[live] for v in (bx, by, k, gx, gy, s))
    assert live.sum() > n // 2
    assert same(gpu.scalar_mult(cid, s, gx, gy), [v[live] for v in B])
    exp = oracle.scalar_mult(oid, k, bx, by, threads=THREADS)
    for fl in (0, LADDER_RADIX32):
        got = [engine.to_numpy(t) for t in engine.scalar_mult(cid, engine.to_device(k), engine.to_device(bx), engine.to_device(by), flags=fl)]
        assert same(got, exp), (bits, a_kind, fl, np.flatnonzero((got[0] != exp[0]).any(axis=1))[:6])
    ax, ay = oracle.to_affine(oid, exp)
    gax, gay = gpu.scalar_mult(cid, k, bx, by, affine=True)
    assert np.array_equal(gax, ax) and np.array_equal(gay, ay)
    P = oracle.from_affine(oid, bx, by)
    (R, Pu), (Rg, Pug) = oracle.dblu(oid, P), gpu.dblu(cid, P)
    assert same(Rg + Pug, R + Pu)
    (R3, Pu2), (R3g, Pu2g) = oracle.zaddu(oid, Pu, R), gpu.zaddu(cid, Pu, R)
    assert same(R3g + Pu2g, R3 + Pu2)
    assert same(gpu.add_z2_1(cid, R3, (P[0], P[1])), oracle.add_z2_1(oid, R3, (P[0], P[1])))
    on = engine.to_numpy(engine.on_curve(cid, engine.to_device(bx), engine.to_device(by)))
    assert on.all()
    xs = np.concatenate([bx, ints_to_arr([(v * 7 + 3) % c["p"] for v in range(64)])])           # x of points, and x values of which about half have no point
    (yg, okg), (yo, oko) = gpu.compute_y(cid, xs), oracle.compute_y(oid, xs)                       # the square root: 29-bit sliding windows against the oracle's power ladder
    assert np.array_equal(okg, oko) and np.array_equal(yg, yo) and okg[:len(bx)].all() and (bits < 20 or not okg.all())
    off = by.copy(); off[:, 0] ^= np.uint64(1)
    assert not engine.to_numpy(engine.on_curve(cid, engine.to_device(bx), engine.to_device(off))).any()
    from ecsimd_amd import EcsimdHipError, ALG_WINDOWED
    kd = engine.to_device(k)
    with pytest.raises(EcsimdHipError, match="group order"):
        engine.scalar_mult_base(cid, kd, flags=OUT_AFFINE | ALG_WINDOWED)
    with pytest.raises(EcsimdHipError, match="group order"):
        engine.ecdsa_verify(cid, kd, kd, kd, kd, kd)
    lx, ly = engine.scalar_mult_base(cid, kd, flags=OUT_AFFINE)                                   # no order: the small-batch route does not apply, the ladder runs
    ex = oracle.to_affine(oid, oracle.scalar_mult(oid, k, gx, gy, threads=THREADS))
    assert np.array_equal(engine.to_numpy(lx), ex[0]) and np.array_equal(engine.to_numpy(ly), ex[1])
