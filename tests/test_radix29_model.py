"""The reduced-radix (9 x 29-bit signed limbs, R' = 2^261) field arithmetic of ecsimd_amd/csrc/fe29.cuh, on the CPU:
  * the interval proof that one ladder iteration cannot overflow a 32-bit limb or a 64-bit column and maps the loop invariant into itself;
  * the exact model against the big-int ZDAU (curve_group.h:120-153) on random, extreme and adversarial loop states.
The kernel itself is compared with the oracle on the GPU (tests/test_gpu_parity.py)."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import radix29_model as m  # noqa: E402

CURVES = [m.CURVE_P256, m.CURVE_SECP]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_one_iteration_maps_the_invariant_into_itself_without_overflow(cv):
    r = m.prove_invariant(cv)
    assert r["worst_column_bits"] <= 63 and r["worst_limb_bits"] <= 31


def _state(cv, rng, x1, y1, x2, y2, z, lazy):
    """A loop state holding the field elements x1 .. z (Montgomery form, R' = 2^261), tight or at the lazy end of the invariant."""
    p = cv.p; R = 1 << m.RBITS
    mont = lambda v: m.to_limbs(v * R % p)
    E = m.Exact(cv)
    if not lazy:
        X1, X2, Y1, Y2 = mont(x1), mont(x2), mont(y1), mont(y2)
    else:
        # the shapes the loop really produces: x = D - (W1 + W2), y = P - A with tight D, W, P, A chosen at random
        def split3(v):
            a, b = rng.randrange(p), rng.randrange(p)
            return E.sub(mont((v + a + b) % p), E.add(mont(a), mont(b)))
        def split2(v):
            a = rng.randrange(p)
            return E.sub(mont((v + a) % p), mont(a))
        X1, X2, Y1, Y2 = split3(x1), split3(x2), split2(y1), split2(y2)
    # dx = x1 - x2 as a difference of tight values, dy likewise
    a = rng.randrange(p)
    dx = E.sub(mont((x1 - x2 + a) % p), mont(a))
    dy = E.sub(mont((y1 - y2 + a) % p), mont(a))
    return {"x1": X1, "x2": X2, "dx": dx, "y1": Y1, "dy": dy, "z": mont(z)}


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_exact_model_equals_the_big_int_zdau(cv):
    rng = random.Random(29)
    p = cv.p; R = 1 << m.RBITS; Rinv = pow(R, -1, p)
    E = m.Exact(cv)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 255) % p, (1 << 232) - 1, (1 << 232), m.M29, int("1fffffff" * 8, 16) % p]
    val = lambda l: m.from_limbs(l) * Rinv % p
    for it in range(300):
        pick = (lambda: rng.choice(edge)) if it < 100 else (lambda: rng.randrange(p))
        x1, y1, x2, y2, z = (pick() for _ in range(5))
        st = _state(cv, rng, x1, y1, x2, y2, z, lazy=bool(it & 1))
        sw = bool(rng.getrandbits(1))
        # several iterations in a row: the state stays inside the machine limits (asserted in Exact) and tracks the big-int values
        for _ in range(3):
            out = m.zdau29(E, st, sw)
            e = m.zdau_field(p, x1, y1, x2, y2, z)
            if sw:
                e = (e[2], e[3], e[0], e[1], e[4])
            got = (val(out["x1"]), val(out["y1"]), val(out["x2"]), val(E.sub(out["y1"], out["dy"])), val(out["z"]))
            assert got == e
            assert val(out["dx"]) == (e[0] - e[2]) % p
            st = out; x1, y1, x2, y2, z = e


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_montgomery_product_and_square_on_extreme_limbs(cv):
    """mul / sqr on limb vectors at the corners of what the loop may hand them (every limb at an interval end)."""
    rng = random.Random(7)
    p = cv.p; R = 1 << m.RBITS
    E = m.Exact(cv)
    B = 1 << m.W
    ends_a = [-2 * B, -B, -1, 0, 1, B - 1]
    ends_b = [-B + 1, -1, 0, 1, B - 1]
    for _ in range(2000):
        a = [rng.choice(ends_a) for _ in range(8)] + [rng.randrange(-(1 << 26), 1 << 26)]
        b = [rng.choice(ends_b) for _ in range(8)] + [rng.randrange(-(1 << 26), 1 << 26)]
        r = E.mul(a, b)
        assert (m.from_limbs(r) * R - m.from_limbs(a) * m.from_limbs(b)) % p == 0
        assert all(0 <= x <= m.M29 for x in r[:8])
        s = E.sqr(b)
        assert (m.from_limbs(s) * R - m.from_limbs(b) ** 2) % p == 0


# ---------------------------------------------------------------- the combs' mixed addition (fe29.cuh madd29)
@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_one_comb_addition_maps_its_invariant_into_itself_without_overflow(cv):
    r = m.prove_comb_invariant(cv)
    assert r["worst_column_bits"] <= 63 and r["worst_limb_bits"] <= 31


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_comb_addition_needs_all_three_carry_passes(cv):
    """Dropping the pass on H overflows a doubled limb on both curves; on P-256 dropping either of the other two overflows a column -- so the
    device function keeps all three on both curves."""
    def variant(nH, nr, nv):
        def f(E, X1, Y1, Z1, x2, y2):
            N = lambda on, v: E.norm(v) if on else v
            Z1Z1 = E.sqr(Z1); U2 = E.mul(x2, Z1Z1); S2 = E.mul(y2, E.mul(Z1Z1, Z1))
            H = N(nH, E.sub(U2, X1)); r = N(nr, E.sub(S2, Y1))
            HH = E.sqr(H); HHH = E.mul(H, HH); V = E.mul(X1, HH)
            X3 = E.sub(E.sub(E.sqr(r), HHH), E.dbl(V))
            return X3, E.sub(E.mul(r, N(nv, E.sub(V, X3))), E.mul(Y1, HHH)), E.mul(Z1, H)
        return f
    inv = m.comb_invariant(cv)
    c = lambda k: m.Iv(inv[k].l, inv[k].v)
    for flags in ((0, 1, 1), (1, 0, 0)) + (((1, 0, 1), (1, 1, 0)) if cv is m.CURVE_P256 else ()):
        with pytest.raises(AssertionError, match="overflow"):
            variant(*flags)(m.Bounds(cv), c("X"), c("Y"), c("Z"), c("tx"), c("ty"))
    variant(1, 1, 1)(m.Bounds(cv), c("X"), c("Y"), c("Z"), c("tx"), c("ty"))


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_exact_comb_additions_equal_the_big_int_formulas(cv):
    """A comb's life on concrete integers with the machine limits asserted: start from a table entry, 64 additions of tight (x, +-y) in a row
    (extreme and random coordinates: they need not be curve points, the formulas are a DAG over GF(p)), against the formulas mod p."""
    rng = random.Random(61)
    p = cv.p; R = 1 << m.RBITS; Rinv = pow(R, -1, p)
    E = m.Exact(cv)
    tight = lambda v: m.to_limbs(v * R % p)
    val = lambda l: m.from_limbs(l) * Rinv % p
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 255) % p, (1 << 232) - 1, 1 << 232, m.M29]
    for trial in range(40):
        pick = (lambda: rng.choice(edge)) if trial < 10 else (lambda: rng.randrange(p))
        x, y = pick(), pick()
        X, Y, Z = tight(x), tight(y), tight(1)
        fx, fy, fz = x, y, 1
        for _ in range(64):
            x2, y2 = pick(), pick()
            neg = rng.getrandbits(1)
            ty = [-v for v in tight(y2)] if neg else tight(y2)
            X, Y, Z = m.madd29(E, X, Y, Z, tight(x2), ty)
            fx, fy, fz = m.madd_field(p, fx, fy, fz, x2, (-y2) % p if neg else y2)
            assert (val(X), val(Y), val(Z)) == (fx, fy, fz)


# ---------------------------------------------------------------- the variable-base window loop (fe29.cuh jdbl29, dbl_add29, vred29)
@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_one_window_of_the_variable_base_loop_maps_its_invariant_into_itself(cv):
    r = m.prove_window_invariant(cv)
    assert r["worst_column_bits"] <= 63 and r["worst_limb_bits"] <= 31


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_exact_window_loop_equals_the_big_int_formulas(cv):
    """63 windows in a row (three doublings and a double-add each) on concrete integers with the machine limits asserted, against the
    Jacobian formulas mod p; coordinates are arbitrary field elements (the formulas are a DAG over GF(p))."""
    rng = random.Random(97)
    p = cv.p; R = 1 << m.RBITS; Rinv = pow(R, -1, p)
    a = -3 if cv is m.CURVE_P256 else 0
    E = m.Exact(cv)
    tight = lambda v: m.to_limbs(v * R % p)
    val = lambda l: m.from_limbs(l) * Rinv % p
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 255) % p, (1 << 232) - 1, 1 << 232, m.M29]
    for trial in range(12):
        pick = (lambda: rng.choice(edge)) if trial < 4 else (lambda: rng.randrange(p))
        fx, fy, fz = pick(), pick(), 1
        X, Y, Z = tight(fx), tight(fy), tight(1)
        for _ in range(63):
            for _ in range(3):
                X, Y, Z = m.jdbl29(E, X, Y, Z)
                fx, fy, fz = m.jdbl_field(p, a, fx, fy, fz)
                assert (val(X), val(Y), val(Z)) == (fx, fy, fz)
            x2, y2 = pick(), pick()
            neg = rng.getrandbits(1)
            ty = [-v for v in tight(y2)] if neg else tight(y2)
            X, Y, Z = m.dbl_add29(E, X, Y, Z, tight(x2), ty)
            fx, fy, fz = m.dbl_add_field(p, fx, fy, fz, x2, (-y2) % p if neg else y2)
            assert (val(X), val(Y), val(Z)) == (fx, fy, fz)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_without_the_value_reduction_the_doublings_grow(cv):
    """Why vred29 exists: on lazy limbs no doubling ever takes a multiple of p away, and the constants 3, 4, 8 push values past the point
    (~10 p) where a Montgomery product stops shrinking them -- the interval run overflows within a window."""
    class NoRed(m.Bounds):
        def vred(self, a):
            return a
    E = NoRed(cv)
    inv = m.window_invariant(cv)
    X, Y, Z = (m.Iv(inv[k].l, inv[k].v) for k in "XYZ")
    with pytest.raises(AssertionError):
        for _ in range(6):
            X, Y, Z = m.jdbl29(E, X, Y, Z)
            assert abs(X.v[0]) < 8 * cv.p and abs(X.v[1]) < 8 * cv.p, "value escapes"


# ---------------------------------------------------------------- the complete addition law of secp256k1 (fe29.cuh pdbl29, padd29)
def test_complete_addition_law_keeps_its_invariant_without_overflow():
    r = m.prove_complete_invariant(m.CURVE_SECP)
    assert r["worst_column_bits"] <= 63 and r["worst_limb_bits"] <= 31


def test_exact_complete_addition_law_equals_the_big_int_formulas_and_the_affine_law():
    """33 windows of four doublings and two additions from the neutral element (0 : 1 : 0), on integers with the machine limits asserted:
    the projective triple equals the formulas mod p at every step, and -- with real curve points -- the affine group law, P + P, P + (-P) = O
    and O + T included (the completeness the constant-time GLV loop relies on)."""
    cv = m.CURVE_SECP
    rng = random.Random(21)
    p = cv.p; R = 1 << m.RBITS; Rinv = pow(R, -1, p)
    E = m.Exact(cv)
    tight = lambda v: m.to_limbs(v * R % p)
    val = lambda l: m.from_limbs(l) * Rinv % p
    X, Y, Z = tight(0), tight(1), tight(0)
    f = (0, 1, 0)
    for w in range(33):
        for _ in range(4):
            X, Y, Z = m.pdbl29(E, X, Y, Z); f = m.pdbl_field(p, *f)
            assert (val(X), val(Y), val(Z)) == f
        for _ in range(2):
            x2, y2 = rng.randrange(p), rng.randrange(p)
            neg = rng.getrandbits(1)
            ty = [-v for v in tight(y2)] if neg else tight(y2)
            X, Y, Z = m.padd29(E, X, Y, Z, tight(x2), ty); f = m.padd_field(p, *f, x2, (-y2) % p if neg else y2)
            assert (val(X), val(Y), val(Z)) == f
    # the group law on real points of y^2 = x^3 + 7
    G = (0x79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798, 0x483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8)
    def aff(P):
        x, y, z = (val(c) for c in P)
        return None if z == 0 else (x * pow(z, -1, p) % p, y * pow(z, -1, p) % p)
    def add_aff(P, Q):
        if P is None: return Q
        if Q is None: return P
        if P[0] == Q[0]:
            if (P[1] + Q[1]) % p == 0: return None
            lam = 3 * P[0] * P[0] * pow(2 * P[1], -1, p) % p
        else:
            lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
        x3 = (lam * lam - P[0] - Q[0]) % p
        return x3, (lam * (P[0] - x3) - P[1]) % p
    O = (tight(0), tight(1), tight(0))
    P1 = m.padd29(E, *O, tight(G[0]), tight(G[1]))                                  # O + G
    assert aff(P1) == G
    P2 = m.padd29(E, *P1, tight(G[0]), tight(G[1]))                                 # G + G through the ADDITION formula
    assert aff(P2) == add_aff(G, G) == aff(m.pdbl29(E, *P1))
    assert aff(m.padd29(E, *P1, tight(G[0]), [-v for v in tight(G[1])])) is None    # G + (-G) = O
    assert aff(m.pdbl29(E, *O)) is None                                             # 2 O = O
    acc, ref = P2, add_aff(G, G)
    for _ in range(20):
        acc = m.padd29(E, *m.pdbl29(E, *acc), tight(G[0]), tight(G[1])); ref = add_aff(add_aff(ref, ref), G)
        assert aff(acc) == ref


# ---------------------------------------------------------------- the default GLV loop of secp256k1 (fe29.cuh madd29_hr, madd29v_finish, is_zero29)
def test_default_glv_loop_keeps_the_window_invariant_without_overflow():
    r = m.prove_glv_invariant(m.CURVE_SECP)
    assert r["worst_column_bits"] <= 63 and r["worst_limb_bits"] <= 31


def test_the_extra_carry_pass_of_madd29v_is_needed():
    """Without the carry pass on X3 the sum leaves the window invariant (X3's limbs reach -3.x 2^29), which jdbl29 was proven on."""
    cv = m.CURVE_SECP
    inv = m.window_invariant(cv)
    E = m.Bounds(cv)
    X3, Y3, Z3 = m.madd29(E, *(m.Iv(inv[k].l, inv[k].v) for k in ("X", "Y", "Z", "tx", "ty")))
    assert not E.vred(X3).within(inv["X"]) and E.vred(E.norm(X3)).within(inv["X"])


def test_exact_default_glv_loop_equals_the_big_int_formulas_and_is_zero29_decides_field_zero():
    """33 windows of four jdbl29 and two madd29v on integers with the machine limits asserted, against the formulas mod p; then is_zero29 on
    the H and r of additions that DO meet R = +-T (lazy accumulator against the tight table point of the same field value) and of ones that do not."""
    cv = m.CURVE_SECP
    rng = random.Random(33)
    p = cv.p; R = 1 << m.RBITS; Rinv = pow(R, -1, p)
    E = m.Exact(cv)
    tight = lambda v: m.to_limbs(v * R % p)
    val = lambda l: m.from_limbs(l) * Rinv % p
    beta = 0x7ae96a2b657c07106e64479eac3434e99cf0497512f58995c1396c28719501ee
    fx, fy, fz = rng.randrange(p), rng.randrange(p), 1
    X, Y, Z = tight(fx), tight(fy), tight(1)
    zeros = 0
    for w in range(33):
        for _ in range(4):
            X, Y, Z = m.jdbl29(E, X, Y, Z); fx, fy, fz = m.jdbl_field(p, 0, fx, fy, fz)
            assert (val(X), val(Y), val(Z)) == (fx, fy, fz)
        for lam in (False, True):
            zi = pow(fz, -1, p)
            case = rng.randrange(4) if fz else 3
            if case == 0:   x2, y2 = fx * zi * zi % p, fy * zi ** 3 % p                  # T = R: H = 0, r = 0
            elif case == 1: x2, y2 = fx * zi * zi % p, (-fy * zi ** 3) % p               # T = -R: H = 0, r != 0
            else:           x2, y2 = rng.randrange(p), rng.randrange(p)
            tx = E.mul(tight(x2 * pow(beta, -1, p) % p), tight(beta)) if lam else tight(x2)
            assert val(tx) == x2
            neg = rng.getrandbits(1)
            ty = [-v for v in tight((-y2) % p)] if neg else tight(y2)
            X3, Y3, Z3, H, r = m.madd29v(E, X, Y, Z, tx, ty, with_hr=True)
            hz, rz = m.is_zero29(E, list(H)), m.is_zero29(E, list(r))
            assert hz == (case in (0, 1)) and rz == (case == 0 or (y2 * fz ** 3 - fy) % p == 0), (w, case)
            zeros += hz
            if hz:                                                                        # the kernel takes the tangent / infinity instead; go on from T's double
                X, Y, Z = m.jdbl29(E, tx, ty, tight(1)); fx, fy, fz = m.jdbl_field(p, 0, x2, y2, 1)
            else:
                X, Y, Z = X3, Y3, Z3; fx, fy, fz = m.madd_field(p, fx, fy, fz, x2, y2)
            assert (val(X), val(Y), val(Z)) == (fx, fy, fz)
    assert zeros >= 10


def test_beta_constant_of_the_glv_loops():
    """k_varwin.inc glv_consts::BETA29 is beta * 2^261 mod p in tight limbs, beta the cube root of unity of BETA (same file)."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ecsimd_amd", "csrc", "k_varwin.inc")).read()
    words = [int(w, 16) for w in re.findall(r"0x[0-9a-f]+", re.search(r"BETA\[8\] = \{([^}]*)\}", src).group(1))]
    limbs = [int(w, 16) for w in re.findall(r"0x[0-9a-f]+", re.search(r"BETA29\[9\] = \{([^}]*)\}", src).group(1))]
    beta = sum(w << (32 * i) for i, w in enumerate(words))
    p = m.CURVE_SECP.p
    assert pow(beta, 3, p) == 1 and beta != 1
    assert limbs == m.to_limbs(beta * (1 << m.RBITS) % p)


def test_exact_isomorphic_table_of_the_glv_loop():
    """k_varwin.inc k_varwin_table_iso on integers with the machine limits asserted: 2P by jdbl29, (k + 1)P = P + kP by zaddu29 (co-Z),
    then the backward walk with f_k = dx_k .. dx_7.  Every entry must be (x_k Zg^2, y_k Zg^3) for the AFFINE k P = (x_k, y_k) of the curve and
    Zg = Z_8; and a window of the loop run on those entries must give, with Z' Zg, the point the same window gives on the curve itself."""
    cv = m.CURVE_SECP
    p = cv.p; R = 1 << m.RBITS; Rinv = pow(R, -1, p)
    E = m.Exact(cv)
    tight = lambda v: m.to_limbs(v * R % p)
    val = lambda l: m.from_limbs(l) * Rinv % p
    canon = lambda l: m.to_limbs(m.from_limbs(l) % p)                                     # canon29 + to29: the canonical residue, tight limbs
    G = (0x79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798, 0x483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8)

    def add_aff(P, Q):
        if P[0] == Q[0]:
            lam = 3 * P[0] * P[0] * pow(2 * P[1], -1, p) % p
        else:
            lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
        x3 = (lam * lam - P[0] - Q[0]) % p
        return x3, (lam * (P[0] - x3) - P[1]) % p
    rng = random.Random(5)
    P = G
    below = False
    for trial in range(24):
        for _ in range(rng.randrange(1, 40)):
            P = add_aff(P, G) if P != G else add_aff(G, G)
        x1, y1 = tight(P[0]), tight(P[1])
        slots = {}
        h = [None] * 7
        X, Y, Z = m.jdbl29(E, x1, y1, tight(1))
        h[0] = Z
        zz = E.sqr(Z)
        qx, qy = E.mul(x1, zz), E.mul(y1, E.mul(zz, Z))                                   # P over Z_2
        for k in range(2, 8):
            slots[k] = (X, Y)                                                             # as born (scratch), lazy limbs and all
            X, Y, qx, qy, Z, dx = m.zaddu29(E, qx, qy, X, Y, Z)
            h[k - 1] = dx
        for v in (E.vred(X), E.vred(Y), Z):                                              # what the kernel hands canon29 (its domain: (-p, 2p))
            assert -p < m.from_limbs(v) < 2 * p
        below |= m.from_limbs(X) <= -p
        slots[8] = (canon(X), canon(Y))
        zg = canon(Z)
        f = h[6]
        for k in range(7, 0, -1):
            Xk, Yk = (x1, y1) if k == 1 else slots[k]
            f2 = E.sqr(f)
            slots[k] = (canon(E.mul(Xk, f2)), canon(E.mul(Yk, E.mul(f2, f))))
            if k > 1:
                f = E.mul(f, h[k - 2])
        Zg = val(zg)
        assert Zg != 0
        kP = P
        for k in range(1, 9):
            if k > 1:
                kP = add_aff(kP, P)
            assert (val(slots[k][0]), val(slots[k][1])) == (kP[0] * Zg * Zg % p, kP[1] * pow(Zg, 3, p) % p), k
        # one window on the isomorphic curve: R = 16 (3P) + 5P - beta-free -- against the affine law
        X, Y, Z = slots[3][0], slots[3][1], tight(1)
        for _ in range(4):
            X, Y, Z = m.jdbl29(E, X, Y, Z)
        X, Y, Z = m.madd29v(E, X, Y, Z, slots[5][0], [-v for v in slots[5][1]])        # - 5P
        zt = val(Z) * Zg % p                                                              # Z' Zg: the point on the curve itself
        aff = (val(X) * pow(zt, -2, p) % p, val(Y) * pow(zt, -3, p) % p)
        want = P
        for _ in range(42):                                                               # 16 * 3 - 5 = 43
            want = add_aff(want, P)
        assert aff == want
    assert below                                                                          # the case vred29 is there for did occur


def test_isomorphic_table_walk_stays_inside_the_machine_and_canon29s_domain():
    r = m.prove_iso_table(m.CURVE_SECP)
    assert r["worst_column_bits"] <= 63 and r["worst_limb_bits"] <= 31


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_exact_square_root_chains_on_29_bit_limbs(cv):
    """point.cuh fe_sqrt_candidate29: the addition chains for a^((p+1)/4) with sqr29 / mul29 on tight operands, machine limits asserted, against pow()."""
    p = cv.p; R = 1 << m.RBITS; Rinv = pow(R, -1, p)
    E = m.Exact(cv)
    tight = lambda v: m.to_limbs(v * R % p)
    val = lambda l: m.from_limbs(l) * Rinv % p
    def sqr_n(a, n):
        for _ in range(n):
            a = E.sqr(a)
        return a
    rng = random.Random(77)
    for xv in [0, 1, 2, p - 1, p - 2, (p - 1) // 2, m.M29, (1 << 232) - 1] + [rng.randrange(p) for _ in range(6)]:
        x = tight(xv)
        x2 = E.mul(E.sqr(x), x)
        if cv is m.CURVE_P256:
            x4 = E.mul(sqr_n(x2, 2), x2); x8 = E.mul(sqr_n(x4, 4), x4); x16 = E.mul(sqr_n(x8, 8), x8); x32 = E.mul(sqr_n(x16, 16), x16)
            t = E.mul(sqr_n(x32, 32), x); t = E.mul(sqr_n(t, 96), x); t = sqr_n(t, 94)
        else:
            x3 = E.mul(E.sqr(x2), x); x6 = E.mul(sqr_n(x3, 3), x3); x9 = E.mul(sqr_n(x6, 3), x3); x11 = E.mul(sqr_n(x9, 2), x2)
            x22 = E.mul(sqr_n(x11, 11), x11); x44 = E.mul(sqr_n(x22, 22), x22); x88 = E.mul(sqr_n(x44, 44), x44); x176 = E.mul(sqr_n(x88, 88), x88)
            x220 = E.mul(sqr_n(x176, 44), x44); x223 = E.mul(sqr_n(x220, 3), x3)
            t = E.mul(sqr_n(x223, 23), x22); t = E.mul(sqr_n(t, 6), x2); t = sqr_n(t, 2)
        assert val(t) == pow(xv, (p + 1) // 4, p), hex(xv)


# ---------------------------------------------------------------- round 5: the proofs are about THE DEVICE CODE
import re  # noqa: E402

import fe29_structure as fs  # noqa: E402


@pytest.mark.parametrize("fn,kind", fs.COVERED, ids=[f"{f.__name__}-{k}" for f, k in fs.COVERED])
def test_the_device_functions_are_the_model_functions_node_for_node(fn, kind):
    """fe29.cuh parsed into the expression DAG of each function's outputs == the DAG the model function builds on symbols: the same products and squares on
    the same operands, the same carry passes (norm29, with their shifts) and value reductions (vred29) in the same places, the same per-curve variants."""
    dev, mod = fn(kind)
    assert set(dev) == set(mod)
    for k in dev:
        assert dev[k] == mod[k], (fn.__name__, kind, k)


def test_the_any_prime_ladder_invariant_and_dense_exact_model():
    """Round 5 (curves registered at run time): the ladder's invariant holds for EVERY odd p < 2^256 at once (every limb of p an interval), and the exact
    model with a dense reduction -- q_k = column * (-p^-1), then q_k p over all nine limbs -- tracks the big-int ZDAU on three real primes."""
    r = m.prove_invariant(m.CURVE_ANY)
    assert r["worst_column_bits"] <= 63 and r["worst_limb_bits"] <= 31
    rng = random.Random(5)
    for name, p in (("brainpoolP256r1", m.BRAINPOOL_P256), ("sm2", m.SM2_P), ("frp256v1", m.FRP256_P), ("p256 as a dense prime", m.P256), ("a 192-bit prime", 2**192 - 2**64 - 1), ("7", 7)):
        cv = m.Curve.dense(name, p)
        E = m.Exact(cv)
        R = 1 << m.RBITS; Rinv = pow(R, -1, p)
        val = lambda l: m.from_limbs(l) * Rinv % p
        for it in range(40):
            x1, y1, x2, y2, z = (rng.randrange(p) for _ in range(5))
            st = _state(cv, rng, x1, y1, x2, y2, z, lazy=bool(it & 1))
            for _ in range(3):
                out = m.zdau29(E, st, False)
                exp = m.zdau_field(p, x1, y1, x2, y2, z)
                got = (val(out["x1"]), val(out["y1"]), val(out["x2"]), (val(out["y1"]) - val(out["dy"])) % p, val(out["z"]))
                assert got == exp, name
                assert val(out["dx"]) == (exp[0] - exp[2]) % p
                st = out; x1, y1, x2, y2, z = exp


COVERED_FUNCTIONS = ["zdau29", "madd29_hr", "madd29_finish", "madd29v_finish", "jdbl29", "dbl_add29", "gjdbl29", "zaddu29", "mul21_29", "pdbl29", "padd29"]


def _call_sites(text):
    """(start, end, inner) of every norm29 / norm29<S> / vred29<C> call inside the covered functions of fe29.cuh."""
    sites = []
    for name in COVERED_FUNCTIONS:
        mo = re.search(r"ECS_DEV\s+[\w:<>]+\s+" + name + r"\s*\(", text)
        k = text.index("{", text.index(")", mo.end()))
        depth, e = 1, k + 1
        while depth:
            depth += {"{": 1, "}": -1}.get(text[e], 0); e += 1
        for c in re.finditer(r"\b(norm29(?:<\d>)?|vred29<C>)\(", text[k:e]):
            s0 = k + c.start(); i = k + c.end(); d = 1
            while d:
                d += {"(": 1, ")": -1}.get(text[i], 0); i += 1
            sites.append((s0, i, text[k + c.end():i - 1], name, c.group(1)))
    return sites


def test_dropping_any_single_carry_pass_or_value_reduction_from_the_device_is_detected():
    """The mutation the structural check exists for: delete ONE norm29 / vred29 call from fe29.cuh (keep its operand) -- for every one of them some
    covered (function, curve) pair must stop matching the model.  (Round 4 pruned eleven carry passes by search; a twelfth dropped from the device alone
    would have passed every other test in the tree.)"""
    text = fs._strip_comments(open(fs.FE29).read())
    sites = _call_sites(text)
    assert len(sites) >= 40, len(sites)
    # a shifting pass is not removable syntactically (norm29<2>(x) = 4x): turn it into the plain pass instead -- the model must notice the shift too
    for s0, e, inner, name, what in sites:
        mutated = text[:s0] + ("norm29(" + inner + ")" if what.startswith("norm29<") else inner) + text[e:]
        caught = False
        for fn, kind in fs.COVERED:
            try:
                dev, mod = fn(kind, mutated)
            except (KeyError, ValueError):
                caught = True; break
            if dev != mod:
                caught = True; break
        assert caught, (name, what, inner)


def test_the_witness_fixture_is_the_exact_models_and_reaches_the_proven_bounds():
    """tests/golden/fe29_witnesses.json (run on the device by tests/test_gpu_witness.py): every entry re-executed on the exact model gives
    the recorded outputs and the recorded worst column, no machine limit is crossed, and the product-level entries sit where they claim: for every proof on the
    two built-in primes some operand pair reaches >= 15/16 of the column bound the proof derives for that call -- the largest of all exactly 2^63 - 2^34."""
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fe29_witnesses.json")
    data = json.load(open(path))
    curves = {"p256": m.CURVE_P256, "secp256k1": m.CURVE_SECP}
    curves.update({k: m.Curve.dense(k, int(v, 16)) for k, v in data["curves"].items()})
    best = {}
    for e in data["entries"]:
        E = m.Exact(curves[e["curve"]])
        if e["kind"] == "product":
            out = [E.mul(e["in"][0], e["in"][1]) if e["op"] == "mul" else E.sqr(e["in"][0])]
            key = (e["proof"], e["curve"])
            best[key] = max(best.get(key, 0), e["worst_column"] / e["proven_column"])
            assert e["worst_column"] <= e["proven_column"] < 2**63
        else:
            out = m.OPS[e["op"]][3](E, [list(x) for x in e["in"]], bool(e["swap"]))
        assert out == e["out"] and E.worst_col == e["worst_column"], (e["curve"], e["op"])
    for (proof, curve), ratio in best.items():
        if curve in ("p256", "secp256k1"):
            assert ratio >= 15 / 16, (proof, curve, ratio)
    assert max(e["worst_column"] for e in data["entries"]) >= 2**63 - 2**35
    import subprocess
    assert subprocess.run([sys.executable, os.path.join(os.path.dirname(path), "..", "..", "tools", "make_witnesses.py"), "--check"], capture_output=True).returncode == 0


def test_power_chains_need_no_carry_pass():
    """gcurve.cuh gc_pow29 (the square root on a registered curve: sliding windows over a, a^3, a^5, a^7) and point.cuh fe_sqrt_candidate29 (the built-in primes'
    fixed chains) multiply and square nothing but outputs of products: that set is closed under sqr29 / mul29 -- for P-256, secp256k1 and EVERY odd p < 2^256 --
    with the worst column a bit below the ladder's, and its members are in leave29's domain."""
    for cv in (m.CURVE_P256, m.CURVE_SECP, m.CURVE_ANY):
        r = m.prove_pow_chain(cv)
        assert r["worst_column_bits"] <= 62 and r["worst_limb_bits"] <= 30, (cv, r)


# ---------------------------------------------------------------- round 5: the window loop of a registered curve (fe29.cuh gjdbl29; k_gvarwin.hip)
_REGISTERED = [("brainpoolP256r1", m.BRAINPOOL_P256), ("sm2", m.SM2_P), ("frp256v1", m.FRP256_P)]


def test_the_window_loop_and_table_of_a_registered_curve_close_for_every_odd_prime():
    """Modified Jacobian doublings with W = a Z^4 carried, the fused double-add and the co-Z chain of the table, on intervals with every limb of p an
    interval: no overflow, every accumulator inside the invariant again, everything canonicalised inside canon29's domain -- WITHOUT a value reduction."""
    for cv in [m.CURVE_ANY] + [m.Curve.dense(name, p) for name, p in _REGISTERED]:
        for prove in (m.prove_gwindow_invariant, m.prove_gtable):
            r = prove(cv)
            assert r["worst_column_bits"] <= 63 and r["worst_limb_bits"] <= 31


def test_eight_times_a_product_is_what_would_not_close():
    """Why gjdbl29 forms 8 Y^4 as 2 (2 YY)^2: with 8 x (YY^2) -- P-256's form, whose growth vred29 undoes -- the doublings leave the invariant on a dense prime."""
    def grow(E, X, Y, Z, Wc):
        Yn = E.norm(Y); YY = E.sqr(Yn); G = E.norm(YY, 2); B = E.mul(X, G); XX = E.sqr(E.norm(X))
        alpha = E.norm(E.add(E.add(E.dbl(XX), XX), Wc))
        X3 = E.sub(E.sqr(alpha), E.dbl(B))
        E8 = E.dbl(E.norm(E.sqr(YY), 2))
        return X3, E.sub(E.mul(alpha, E.norm(E.sub(B, X3))), E8), E.mul(E.dbl(Yn), Z), Wc
    E = m.Bounds(m.CURVE_ANY)
    inv = m.gwindow_invariant(m.CURVE_ANY)
    X, Y, Z, Wc = (m.Iv(inv[k].l, inv[k].v) for k in "XYZW")
    with pytest.raises(AssertionError):
        for _ in range(8):
            X, Y, Z, Wc = grow(E, X, Y, Z, Wc)
            assert Y.within(inv["Y"]), "value escapes"


@pytest.mark.parametrize("name,p", _REGISTERED + [("p256 as a dense prime", m.P256)], ids=lambda v: v if isinstance(v, str) else "")
def test_exact_window_loop_of_a_registered_curve_equals_the_big_int_formulas(name, p):
    """63 windows (W = a Z^4 from Z, two doublings that carry it, one that does not, a double-add) on integers with the machine limits asserted, against
    the Jacobian formulas mod p for a random coefficient a."""
    rng = random.Random(131)
    cv = m.Curve.dense(name, p)
    R = 1 << m.RBITS; Rinv = pow(R, -1, p)
    E = m.Exact(cv)
    tight = lambda v: m.to_limbs(v * R % p)
    val = lambda l: m.from_limbs(l) * Rinv % p
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 255) % p, (1 << 232) - 1, 1 << 232, m.M29]
    for trial in range(6):
        pick = (lambda: rng.choice(edge)) if trial < 2 else (lambda: rng.randrange(p))
        a = (p - 3, 0, rng.randrange(p), rng.randrange(p), p - 1, 1)[trial]
        ap = tight(a)
        fx, fy, fz = pick(), pick(), 1
        X, Y, Z = tight(fx), tight(fy), tight(1)
        for _ in range(63):
            Wc = m.gw_of_z(E, Z, ap)
            assert val(Wc) == a * pow(fz, 4, p) % p
            for wout in (True, True, False):
                X, Y, Z, Wc = m.gjdbl29(E, X, Y, Z, Wc, wout)
                fx, fy, fz = m.jdbl_field(p, a, fx, fy, fz)
                assert (val(X), val(Y), val(Z)) == (fx, fy, fz)
                if wout:
                    assert val(Wc) == a * pow(fz, 4, p) % p
            x2, y2 = pick(), pick()
            neg = rng.getrandbits(1)
            ty = [-v for v in tight(y2)] if neg else tight(y2)
            X, Y, Z = m.dbl_add29(E, X, Y, Z, tight(x2), ty)
            fx, fy, fz = m.dbl_add_field(p, fx, fy, fz, x2, (-y2) % p if neg else y2)
            assert (val(X), val(Y), val(Z)) == (fx, fy, fz)


def test_exact_table_of_odd_multiples_over_one_z_on_brainpoolP256r1():
    """k_gvarwin.hip k_gvw_table on integers with the machine limits asserted: 2P by gjdbl29, (2j + 3) P = 2P + (2j + 1) P by zaddu29, the walk back.  Every
    entry must be (x_j Zg^2, y_j Zg^3) for the AFFINE (2j + 1) P of the curve; and a window run on those entries with a' = a Zg^4 must give, with Z' Zg, the
    point the affine law gives on the curve itself."""
    from ecsimd_amd.curves import NAMED
    c = NAMED["brainpoolP256r1"]
    p, a = c["p"], c["a"]
    cv = m.Curve.dense("brainpoolP256r1", p)
    R = 1 << m.RBITS; Rinv = pow(R, -1, p)
    E = m.Exact(cv)
    tight = lambda v: m.to_limbs(v * R % p)
    val = lambda l: m.from_limbs(l) * Rinv % p
    canon = lambda l: m.to_limbs(m.from_limbs(l) % p)

    def add_aff(P, Q):
        if P[0] == Q[0]:
            lam = (3 * P[0] * P[0] + a) * pow(2 * P[1], -1, p) % p
        else:
            lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
        x3 = (lam * lam - P[0] - Q[0]) % p
        return x3, (lam * (P[0] - x3) - P[1]) % p
    G = (c["gx"], c["gy"])
    rng = random.Random(9)
    P = G
    for trial in range(12):
        for _ in range(rng.randrange(1, 30)):
            P = add_aff(P, G)
        x1, y1, one = tight(P[0]), tight(P[1]), tight(1)
        X2, Y2, z, _ = m.gjdbl29(E, x1, y1, one, tight(a), False)
        zz = E.sqr(z)
        ax, ay = E.mul(x1, zz), E.mul(y1, E.mul(zz, z))
        dx2, dy2 = X2, Y2
        ex, ey, h = [], [], []
        for j in range(7):
            ex.append(ax); ey.append(ay)
            rx, ry, dx2, dy2, z, dx = m.zaddu29(E, dx2, dy2, ax, ay, z)
            ax, ay = rx, ry
            h.append(dx)
        slots = [None] * 8
        for v in (E.mul(ax, one), E.mul(ay, one), z):
            assert -p < m.from_limbs(v) < 2 * p                                          # canon29's domain
        slots[7] = (canon(E.mul(ax, one)), canon(E.mul(ay, one)))
        zg = canon(z)
        f = h[6]
        for j in range(6, -1, -1):
            f2 = E.sqr(f)
            vx, vy = E.mul(ex[j], f2), E.mul(ey[j], E.mul(f2, f))
            assert -p < m.from_limbs(vx) < 2 * p and -p < m.from_limbs(vy) < 2 * p
            slots[j] = (canon(vx), canon(vy))
            if j > 0:
                f = E.mul(f, h[j - 1])
        Zg = val(zg)
        assert Zg != 0
        twoP = add_aff(P, P)
        kP = P
        for j in range(8):
            if j > 0:
                kP = add_aff(kP, twoP)
            assert (val(slots[j][0]), val(slots[j][1])) == (kP[0] * Zg * Zg % p, kP[1] * pow(Zg, 3, p) % p), j
        # one window on the isomorphic curve: 16 (3P) - 5P = 43 P
        ap = E.mul(tight(a), E.sqr(E.sqr(zg)))
        X, Y, Z = slots[1][0], slots[1][1], one
        Wc = m.gw_of_z(E, Z, ap)
        for wout in (True, True, False):
            X, Y, Z, Wc = m.gjdbl29(E, X, Y, Z, Wc, wout)
        X, Y, Z = m.dbl_add29(E, X, Y, Z, slots[2][0], [-v for v in slots[2][1]])
        zt = val(Z) * Zg % p
        aff = (val(X) * pow(zt, -2, p) % p, val(Y) * pow(zt, -3, p) % p)
        want = P
        for _ in range(42):
            want = add_aff(want, P)
        assert aff == want
