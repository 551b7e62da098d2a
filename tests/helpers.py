"""Shared test helpers: hex <-> limb arrays, the synthetic-input generator, and an independent
big-int model of the curve arithmetic (plain Python ints, textbook affine formulas) used as a
third opinion next to the C oracle and the reference-generated fixtures."""
import numpy as np

from oracle.loader import P256, SECP256K1, from_int, to_int, ints_to_arr, arr_to_ints, from_hex  # noqa: F401

MASK64 = (1 << 64) - 1
SEED = 0x5EEDEC51D0000001

CURVE_PARAMS = {
    P256: dict(p=0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff, a=-3,
               b=0x5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b,
               gx=0x6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296,
               gy=0x4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5,
               n=0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551),
    SECP256K1: dict(p=0xfffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f, a=0, b=7,
                    gx=0x79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798,
                    gy=0x483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8,
                    n=0xfffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141),
}
CURVE_NAMES = {P256: "p256", SECP256K1: "secp256k1"}
R = 1 << 256


def hexes_to_arr(hs, words=4):
    return ints_to_arr([int(h, 16) for h in hs], words)


def arr_to_hexes(a, words=4):
    return [format(v, "0%dx" % (16 * words)) for v in arr_to_ints(a)]


def splitmix64(z):
    z = (z + 0x9e3779b97f4a7c15) & MASK64
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & MASK64
    return z ^ (z >> 31)


def fill_random_np(n, seed, stream, first_index=0, clear_top_bits=0):
    """numpy twin of ecsimd_hip_fill_random (k_bignum.hip k_fill_random)."""
    idx = (np.arange(n, dtype=np.uint64)[:, None] + np.uint64(first_index)) * np.uint64(4) + np.arange(4, dtype=np.uint64)[None, :]
    z = idx ^ np.uint64(seed ^ ((stream << 56) & MASK64))
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9e3779b97f4a7c15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        z = z ^ (z >> np.uint64(31))
    if clear_top_bits:
        z[:, 3] &= np.uint64(MASK64 >> clear_top_bits)
    return z


# ---- independent affine big-int model -------------------------------------------------------
def ec_add(cv, P, Q):
    c = CURVE_PARAMS[cv]; p = c["p"]
    if P is None: return Q
    if Q is None: return P
    x1, y1 = P; x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0: return None
        lam = (3 * x1 * x1 + c["a"]) * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def ec_mul(cv, k, P):
    Rr, Q = None, P
    while k:
        if k & 1: Rr = ec_add(cv, Rr, Q)
        Q = ec_add(cv, Q, Q); k >>= 1
    return Rr


def jacobian_mgry_to_affine_int(cv, X, Y, Z):
    """(X, Y, Z) Montgomery-form ints -> classical affine ints, or None for Z == 0."""
    p = CURVE_PARAMS[cv]["p"]; Rinv = pow(R, -1, p)
    x, y, z = X * Rinv % p, Y * Rinv % p, Z * Rinv % p
    if z == 0: return None
    zi = pow(z, -1, p)
    return x * zi * zi % p, y * zi * zi * zi % p
