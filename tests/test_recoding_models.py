"""CPU suite: integer models of the scalar recodings the windowed kernels use (ecsimd_amd/csrc/k_affine.inc,
k_varwin.inc).  Each model restates the kernel's digit logic in Python and checks the identity it relies on --
sum of digits x weights == the scalar (mod n where the kernel reduces), digit ranges, table sizes -- on edge
patterns and random scalars.  The kernels themselves are checked on the GPU against the ladder, the oracle and
OpenSSL; these tests pin the arithmetic the kernels were written from."""
import random

import pytest

from helpers import CURVE_PARAMS, P256, SECP256K1

M256 = (1 << 256) - 1
EDGE = [0, 1, 2, 7, 8, 9, 15, 16, 0x7f, 0x80, 0x7fff, 0x8000, 0x7ffff, 0x80000, 0x80001, (1 << 128) - 1, 1 << 128, 1 << 255, M256,
        int("8" * 64, 16), int("7" * 64, 16), int("f" * 64, 16), int("80000" * 12, 16), int("7ffff" * 12, 16)]


def scalars(seed, count=2000):
    rng = random.Random(seed)
    return EDGE + [rng.getrandbits(256) for _ in range(count)]


@pytest.mark.parametrize("bits", [6, 7])
def test_signed_window_recoding_of_the_fixed_base_kernels(bits):
    """k_base_windowed_s: u = chunk + carry; u > 2^(b-1) -> digit u - 2^b and carry 1.
    (256 + b) // b windows absorb the final carry; |digit| <= 2^(b-1) = the table's entries per window."""
    windows, half, full = (256 + bits) // bits, 1 << (bits - 1), 1 << bits
    for k in scalars(bits):
        carry, total = 0, 0
        for w in range(windows):
            u = ((k >> (bits * w)) & (full - 1)) + carry
            carry = 1 if u > half else 0
            d = u - full if carry else u
            assert -half <= d <= half
            if bits * w >= 256 - bits and bits * w + bits > 256 and d:                 # the top window never needs more than 2^256 / 2^pos
                assert abs(d) << (bits * w) <= 1 << 256
            total += d << (bits * w)
        assert carry == 0 and total == k


@pytest.mark.parametrize("bits", [16, 18, 20, 22])
@pytest.mark.parametrize("cv", [P256, SECP256K1])
def test_odd_digit_recoding_of_the_device_memory_table_kernel(cv, bits):
    """k_base_windowed_g: k mod n, the odd one of k and n - k; digit i = (bits [b i, b i + b] with bit b i forced to 1) - 2^b
    for all windows but the last, whose digit is the remaining bits | 1.  ceil(256 / b) odd digits, |d| < 2^b, the last
    positive; the table slot of magnitude m is (m - 1) / 2 < 2^(b-1); every needed multiple m 2^(b i) is below 2^256."""
    n = CURVE_PARAMS[cv]["n"]
    windows, full = (256 + bits - 1) // bits, 1 << bits
    for k in scalars(bits + cv, 1000) + [n - 1, n + 1, n - 2, 1, 2]:
        k &= M256
        r = k - n if k >= n else k
        if r == 0:
            continue
        flip = r % 2 == 0
        a = n - r if flip else r
        digits = [(((a >> (bits * w)) & (2 * full - 1)) | 1) - full for w in range(windows - 1)] + [(a >> (bits * (windows - 1))) | 1]
        assert all(d % 2 and abs(d) < full and (abs(d) - 1) // 2 < full // 2 for d in digits) and digits[-1] > 0
        assert all(abs(d) << (bits * w) < 1 << 256 for w, d in enumerate(digits))
        val = sum(d << (bits * w) for w, d in enumerate(digits))
        assert val == a and ((-val if flip else val) - k) % n == 0
        acc = digits[0]                                                             # the accumulator never meets +-(the next entry) or infinity
        for w in range(1, windows):
            t = digits[w] << (bits * w)
            assert acc % n and (acc - t) % n and (acc + t) % n
            acc += t


@pytest.mark.parametrize("cv", [P256, SECP256K1])
def test_odd_digit_recoding_of_the_variable_base_kernel(cv):
    """k_varwin_mult_odd: k <- k mod n; the odd one of k and n - k (n is odd); digit i < 63 = (nibble_i | 1) - 16 when the
    nibble above is even, nibble_i | 1 otherwise; the top digit = top nibble | 1.  Sixty-four odd digits, |d| <= 15,
    magnitude m at table slot (m - 1) / 2; no addition inside the loop meets R = +-T or the point at infinity."""
    n = CURVE_PARAMS[cv]["n"]
    for k in scalars(cv + 300) + [n - 1, n, n + 1, n - 2, (n - 1) // 2, (n + 1) // 2, 1, 2, 3]:
        k &= M256
        r = k - n if k >= n else k
        if r == 0:
            continue                                                                # the kernel returns infinity by flag
        flip = r % 2 == 0
        a = n - r if flip else r
        assert a % 2 == 1 and 0 < a < n
        nibs = [(a >> (4 * j)) & 15 for j in range(64)]
        digits = [(nibs[j] | 1) - (0 if nibs[j + 1] & 1 else 16) for j in range(63)] + [nibs[63] | 1]
        assert all(d % 2 and -15 <= d <= 15 for d in digits) and digits[63] > 0
        assert all(0 <= (abs(d) - 1) // 2 <= 7 for d in digits)
        val = sum(d << (4 * j) for j, d in enumerate(digits))
        assert val == a and ((-val if flip else val) - k) % n == 0
        acc = digits[63]                                                            # discrete log of R: 16 R + d P as 2 (8 R) + d P
        for j in range(62, -1, -1):
            eight = 8 * acc
            assert eight % n and (eight - digits[j]) % n and (eight + digits[j]) % n      # R + T: neither equal nor opposite
            assert (2 * eight + digits[j]) % n and (eight + digits[j] + eight) % n          # (R + T) + R: not opposite; never equal (T != 0)
            acc = 16 * acc + digits[j]
        assert acc == a


def test_glv_split_of_secp256k1():
    """k_varwin_mult_glv: c_i = round(k g_i / 2^384); k1 = k - c1 a1 - c2 a2, k2 = c1 (-b1) - c2 b2 in 256-bit two's
    complement; |k1|, |k2| < 2^128; 33 nibbles after adding 0x888...8 (32 nibbles) with a top digit of 0 or 1."""
    n = CURVE_PARAMS[SECP256K1]["n"]; p = CURVE_PARAMS[SECP256K1]["p"]
    lam = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
    beta = 0x7ae96a2b657c07106e64479eac3434e99cf0497512f58995c1396c28719501ee
    g1 = 0x3086d221a7d46bcde86c90e49284eb153daa8a1471e8ca7fe893209a45dbb031
    g2 = 0xe4437ed6010e88286f547fa90abfe4c4221208ac9df506c61571b4ae8ac47f71
    a1, mb1, a2 = 0x3086d221a7d46bcde86c90e49284eb15, 0xe4437ed6010e88286f547fa90abfe4c3, 0x114ca50f7a8e2f3f657c1108d9d44cfd8
    assert pow(lam, 3, n) == 1 and lam != 1 and pow(beta, 3, p) == 1 and beta != 1
    assert (a1 - mb1 * lam) % n == 0 and (a2 + a1 * lam) % n == 0                  # the lattice basis (a1, b1), (a2, b2 = a1)
    off = int("8" * 32, 16)

    def signed256(v):
        v &= M256
        return v - (1 << 256) if v >> 255 else v
    for k in scalars(7, 5000) + [lam, lam + 1, n - lam, a1, mb1, a2, (a1 * lam) % n, n - 1]:
        k = (k & M256) % n
        c1 = ((k * g1) >> 384) + (((k * g1) >> 383) & 1)
        c2 = ((k * g2) >> 384) + (((k * g2) >> 383) & 1)
        assert c1 < 1 << 128 and c2 < 1 << 128
        k1 = signed256(k - ((c1 * a1) & M256) - ((c2 * a2) & M256))
        k2 = signed256(((c1 * mb1) & M256) - ((c2 * a1) & M256))
        assert (k1 + k2 * lam - k) % n == 0 and abs(k1) < 1 << 128 and abs(k2) < 1 << 128
        s1, s2 = (-1 if k1 < 0 else 1), (-1 if k2 < 0 else 1)
        # k_varwin_mult_glv<WB>: 4-bit windows (8-entry tables) and the 5-bit windows of the 16-entry table over one Z (k_varwin_table_iso)
        for wb, windows in ((4, 33), (5, 26)):
            half = 1 << (wb - 1)
            offw = sum(half << (wb * j) for j in range(windows - 1))
            assert offw == (off if wb == 4 else sum(1 << (5 * t + 4) for t in range(25)))
            both = []
            for v in (abs(k1), abs(k2)):
                u = v + offw
                digits = [((u >> (wb * j)) & (2 * half - 1)) - half for j in range(windows - 1)] + [u >> (wb * (windows - 1))]
                assert 0 <= digits[-1] <= (1 if wb == 4 else 8) and all(-half <= d < half for d in digits[:-1])      # table entries 1 .. half cover every magnitude
                assert sum(d << (wb * j) for j, d in enumerate(digits)) == v
                if wb == 5:                                                         # the kernel moves u up 3 bits and reads bits 128..132: the same digits
                    us = u << 3
                    assert [((us >> (5 * j + 3)) & 31) - 16 for j in range(25)] + [(us >> 128) & 31] == digits and us < 1 << 160
                both.append(digits)
            # the window loop never adds a point to itself or to its opposite (add_checked's branches stay cold)
            acc = 0                                                                 # discrete log of R
            for j in range(windows - 1, -1, -1):
                acc = ((1 << wb) * acc) % n if j != windows - 1 else 0
                for t in (s1 * both[0][j], s2 * both[1][j] * lam):
                    t %= n
                    if t and acc:
                        assert (acc - t) % n and (acc + t) % n
                    acc = (acc + t) % n
            assert acc == k % n
