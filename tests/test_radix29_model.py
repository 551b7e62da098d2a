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
