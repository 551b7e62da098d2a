"""CPU suite: big-int models that WALK THE ACCUMULATOR of every odd-digit window kernel in the kernel's own summation order and assert that no
mixed addition ever meets R = +-T or R = infinity -- except the one addition at the one scalar k* where the kernel substitutes a precomputed
point (k_affine.inc comb_special / comb_take_special, capi.hip ensure_window_table).  This is DESIGN.md section 5's prose argument as a
regression test (the k* defect of rounds 1-2 lived in exactly this gap: the recodings were modelled, the sums were not).

The mixed addition the kernels use (point.cuh madd_hmv / add_z2_1, k_varwin.inc dbl_add) is the chord formula: it is wrong for R = T (needs the
tangent), for R = -T (the sum is infinity) and for R = infinity.  With G of prime order n, "R = c G equals +-T = +-t G" is "c = +-t (mod n)", so
a model over the integers c, t decides it exactly.  Shapes (both curves each):
    top4      k_base_windowed<CT>          64 x 8 entries (2d+1) 16^w G, summed from the TOP window down      (ALG_WINDOWED, BASELINE configs[2])
    bottom5   k_base_windowed_s<5, true>   52 x 16 entries, summed from the bottom                             (ALG_WINDOWED | ALG_CONSTANT_TIME; small batches)
    bottom7   k_base_windowed_s<7, false>  37 x 64 entries, summed from the bottom                             (ALG_WINDOWED_SIGNED)
    bottom20  k_base_windowed_g            13 x 2^19 entries in device memory, summed from the bottom          (ALG_WINDOWED_BIG, u1 G of double_scalar_mult)
    horner4   k_varwin_mult_odd<CT>        per-lane table {1, 3, .., 15} P: three doublings + a fused double-add per window (ALG_WINDOWED, variable base)
Scalars: the 16 797-member digit-pattern family the GPU tests use, every k* and its neighbours / images under k -> n - k and k -> k + n, the
ladder's degenerate scalars, and 10^5 random 256-bit values per curve.  (The GLV loops: k_varwin_mult_glv handles R = +-T behind add_checked's
branch and k_varwin_mult_glv_ct runs the complete addition law -- nothing to exclude; their recoding is modelled in test_recoding_models.py.)
"""
import random

import numpy as np
import pytest

from helpers import CURVE_PARAMS, P256, SECP256K1, arr_to_ints

M256 = (1 << 256) - 1
CURVES = [P256, SECP256K1]


def odd_representative(k, n):
    """The kernels' first step: k mod n (one conditional subtraction: k < 2^256 < 2n), then the odd one of it and n - it.  None for k = 0 mod n
    (the kernels return infinity by mask)."""
    r = k - n if k >= n else k
    if r == 0:
        return None
    return n - r if r % 2 == 0 else r


def kstar(n, bits, windows, from_top):
    """capi.hip ensure_window_table: the one odd scalar whose LAST mixed addition meets R = T.  From the top: n - 2 (n mod 2^bits), and only if
    bit `bits` of it is 0; from the bottom: n - 2 (n mod 2^(bits (windows - 1))).  0 = this kernel has no such scalar."""
    low_bits = bits if from_top else bits * (windows - 1)
    ks = n - 2 * (n % (1 << low_bits))
    if ks <= 0 or (from_top and (ks >> bits) & 1):
        return 0
    return ks


def digits_top4(a):
    nibs = [(a >> (4 * j)) & 15 for j in range(64)]
    return [(nibs[j] | 1) - (0 if nibs[j + 1] & 1 else 16) for j in range(63)] + [nibs[63] | 1]


def digits_bottom(a, bits, windows):
    full = 1 << bits
    return [(((a >> (bits * w)) & (2 * full - 1)) | 1) - full for w in range(windows - 1)] + [(a >> (bits * (windows - 1))) | 1]


def walk_comb(a, n, bits, windows, from_top):
    """[(window, kind)] of the exceptional additions of one comb sum; the sum itself must equal a."""
    d = digits_top4(a) if from_top else digits_bottom(a, bits, windows)
    order = range(windows - 2, -1, -1) if from_top else range(1, windows)
    acc = d[windows - 1] << (bits * (windows - 1)) if from_top else d[0]
    events = []
    for w in order:
        t = d[w] << (bits * w)
        if acc % n == 0:
            events.append((w, "R = infinity"))
        elif (acc - t) % n == 0:
            events.append((w, "R = T"))
        elif (acc + t) % n == 0:
            events.append((w, "R = -T"))
        acc += t
    assert acc == a, (hex(a), bits)
    return events


def walk_horner4(a, n):
    """k_varwin_mult_odd: R = d_63 P; per window 8 R by three doublings, then 2 (8R) + d P as ((8R + dP) + 8R) (k_varwin.inc dbl_add)."""
    d = digits_top4(a)
    acc = d[63]
    events = []
    for j in range(62, -1, -1):
        for _ in range(3):
            if acc % n == 0:
                events.append((j, "doubling infinity"))
            acc *= 2
        t = d[j]
        if acc % n == 0 or (acc - t) % n == 0 or (acc + t) % n == 0:
            events.append((j, "R + T"))
        s = acc + t
        if s % n == 0 or (s - acc) % n == 0 or (s + acc) % n == 0:
            events.append((j, "(R + T) + R"))
        acc = s + acc
    assert acc == a
    return events


SHAPES = {                       # name: (bits, windows, summed from the top)
    "top4": (4, 64, True),
    "bottom5": (5, 52, False),
    "bottom7": (7, 37, False),
    "bottom20": (20, 13, False),
}


def family(cv, randoms):
    from test_oracle import digit_pattern_operands
    n = CURVE_PARAMS[cv]["n"]
    out = arr_to_ints(digit_pattern_operands()[::100])
    for bits, windows, top in SHAPES.values():
        ks = kstar(n, bits, windows, top)
        for v in (ks, n - 2 * (n % (1 << bits)), n - 2 * (n % (1 << (bits * (windows - 1))))):         # the existing and the would-be k* of either order
            out += [v, v + 1, v - 1, v + 2, v - 2, n - v, n - v + 1, n - v - 1, v + n, n - v + n]
    out += [1, 2, 3, n - 1, n - 2, n + 1, n + 2, (n - 1) // 2, (n + 1) // 2, (1 << 256) - n, (1 << 256) - n - 1, (1 << 256) - n + 1, M256, 1 << 255, (1 << 255) - 1]
    rng = random.Random(1000 + cv)
    out += [rng.getrandbits(256) for _ in range(randoms)]
    return [v for v in out if 0 < v <= M256]


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("cv", CURVES)
def test_no_comb_addition_is_exceptional_except_the_substituted_one(cv, shape):
    n = CURVE_PARAMS[cv]["n"]
    bits, windows, top = SHAPES[shape]
    ks = kstar(n, bits, windows, top)
    last = 0 if top else windows - 1
    hits = 0
    for k in family(cv, 100000):
        a = odd_representative(k, n)
        if a is None:
            continue
        ev = walk_comb(a, n, bits, windows, top)
        if a == ks:
            assert ev == [(last, "R = T")], (shape, hex(k), ev)            # the kernel overwrites this lane's sum with the table's k* G
            hits += 1
        else:
            assert ev == [], (shape, hex(k), ev)
    if ks:
        assert hits >= 2                                                   # k* itself and n - k* at least: the substitution path is exercised


@pytest.mark.parametrize("cv", CURVES)
def test_kstar_of_the_shipped_shapes(cv):
    """What ensure_window_table computes, and the two facts its ladder fallback rests on: the 5-bit comb's k* IS one of the reference
    ladder's degenerate scalars (2^256 - n), so its point comes from the ladder on n - k*, which is not."""
    n = CURVE_PARAMS[cv]["n"]
    degenerate = {0, n, n - 1, (1 << 256) - n - 1, (1 << 256) - n}
    assert kstar(n, 5, 52, False) == (1 << 256) - n and kstar(n, 5, 52, False) in degenerate and (n - kstar(n, 5, 52, False)) not in degenerate
    for shape in ("bottom7", "bottom20"):
        bits, windows, top = SHAPES[shape]
        assert kstar(n, bits, windows, top) not in degenerate and kstar(n, bits, windows, top) % 2 == 1
    if cv == P256:
        assert kstar(n, 4, 64, True) == n - 2                              # n = ..2551: n mod 16 = 1, bit 4 of n - 2 clear: k = +-2 (how round 3's test found it)
    else:
        assert kstar(n, 4, 64, True) == 0                                  # n = ..4141: n - 2 has bit 4 set: no such scalar for the 4-bit comb


@pytest.mark.parametrize("cv", CURVES)
def test_no_addition_of_the_variable_base_window_loop_is_exceptional(cv):
    n = CURVE_PARAMS[cv]["n"]
    for k in family(cv, 30000):
        a = odd_representative(k, n)
        if a is not None:
            assert walk_horner4(a, n) == [], hex(k)


def test_the_family_is_the_gpu_tests_family():
    from test_oracle import digit_pattern_operands
    assert len(digit_pattern_operands()[::100]) == 16797
    assert len(family(P256, 10)) > 16797 + 100


@pytest.mark.parametrize("shape", ["top4", "bottom5", "bottom7", "bottom20"])
@pytest.mark.parametrize("name", ["brainpoolP256r1", "sm2", "frp256v1"])
def test_generator_comb_of_a_registered_curve(name, shape):
    """k_gcomb.hip runs the top4 shape (ALG_WINDOWED; small batches), the bottom5 shape (ALG_WINDOWED | ALG_CONSTANT_TIME, ecdsa_sign) the bottom7 shape (ALG_WINDOWED_SIGNED) and the bottom20 shape (ALG_WINDOWED_BIG) with the order of a curve registered at
    run time (n >= 2^255 is what capi.hip ensure_gc_comb asks for): the same walk with that n -- no addition is exceptional except the last one at
    k* = n - 2 (n mod 16) when bit 4 of it is clear (top4) / at k* = n - 2 (n mod 2^252) (bottom7)."""
    from oracle.loader import REF_CURVES
    from test_oracle import digit_pattern_operands
    n = REF_CURVES[name]["n"]
    assert n >> 255 == 1
    bits, windows, top = SHAPES[shape]
    ks = kstar(n, bits, windows, top)
    last = 0 if top else windows - 1
    fam = arr_to_ints(digit_pattern_operands()[::100])
    for v in (ks, n - 2 * (n % 16), n - 2 * (n % (1 << 252)), n - 2 * (n % (1 << 255)), n - 2 * (n % (1 << 240))):
        fam += [v, v + 1, v - 1, v + 2, v - 2, n - v, n - v + 1, n - v - 1, v + n, n - v + n]
    fam += [1, 2, 3, n - 1, n - 2, n + 1, n + 2, (n - 1) // 2, (n + 1) // 2, (1 << 256) - n, (1 << 256) - n - 1, (1 << 256) - n + 1, M256, 1 << 255, (1 << 255) - 1]
    rng = random.Random(sum(name.encode()))
    fam += [rng.getrandbits(256) for _ in range(50000)]
    hits = 0
    for k in fam:
        if not 0 < k <= M256:
            continue
        a = odd_representative(k, n)
        if a is None:
            continue
        ev = walk_comb(a, n, bits, windows, top)
        if a == ks:
            # bottom7: k* = h 2^252 - m (n = h 2^252 + m) has the top digit (h - 1) | 1, which is h only for an odd h -- then the last addition meets R = T; for an
            # even h (brainpoolP256r1: 0xa) NO scalar of this comb is exceptional, and the kernel's substitution of the table's k* G at k* changes nothing
            want = [(last, "R = T")] if (top or (n >> (bits * (windows - 1))) & 1) else []      # (bottom5: the top digit is bit 255 of n: always 1)
            assert ev == want, (name, shape, hex(k), ev)
            hits += 1
        else:
            assert ev == [], (name, shape, hex(k), ev)
    assert hits >= 2 if ks else hits == 0
