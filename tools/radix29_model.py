"""radix29_model.py -- big-int model and overflow proof of the reduced-radix field arithmetic of ecsimd_amd/csrc/fe29.cuh.

A P-256 field element is nine SIGNED 32-bit limbs in radix 2^29 (value = sum l[i] * 2^(29 i)); the Montgomery radix is
R' = 2^261.  Additions and subtractions are limb-wise 32-bit operations without carries, products accumulate in 64-bit
columns (v_mad_i64_i32) without carries between the products of a column, and the Montgomery reduction -- q_k = column mod 2^29
because p = -1 mod 2^29 -- is folded into the same column walk (finely integrated product scanning).

This file is the CPU statement of that arithmetic, function for function (same names, same order of operations as fe29.cuh):

  * `Exact`   -- executes on concrete integers with every machine limit asserted (int32 limbs, int64 columns);
  * `Bounds`  -- executes on INTERVALS (per limb and for the value) and proves that no input inside the loop invariant of the
                 ladder can overflow a limb or a column, and that one ZDAU maps the invariant into itself.

tests/test_radix29_model.py runs both; the GPU tests compare the kernel with the oracle's ZDAU / ladder bit for bit.
Reference formulas: /root/reference/include/ecsimd/curve_group.h:120-153 (ZDAU), mgry_mul.h:84-121 (Montgomery reduction).
"""
from fractions import Fraction

W = 29
NL = 9
M29 = (1 << W) - 1
RBITS = W * NL                      # 261
P256 = 2**256 - 2**224 + 2**192 + 2**96 - 1
SECP = 2**256 - 2**32 - 977
I32 = (-(1 << 31), (1 << 31) - 1)
I64 = (-(1 << 63), (1 << 63) - 1)


class Curve:
    """Sparse signed form of p in radix 2^29: p = sum c * 2^(29 off) over `terms`, with the term at offset 0 equal to -m0inv^-1.
    tight_sq: zdau29 carry-passes dy - u and dx + u before squaring them (fe29.cuh TIGHT_SQ: every reduction but secp256k1's sparse one needs it)."""
    def __init__(self, name, p, terms, qmul, tight_sq=True):
        self.name, self.p, self.terms, self.qmul, self.tight_sq = name, p, terms, qmul, tight_sq
        assert sum(c << (W * o) for o, c in terms) == p, name
        # q = (column * qmul) mod 2^29 makes column + q * (term at offset 0) vanish mod 2^29
        c0 = dict(terms)[0]
        assert (1 + qmul * c0) % (1 << W) == 0, name
        self.term_ranges = [(o, (c, c)) for o, c in terms]     # what the interval execution multiplies the quotient digits by

    @classmethod
    def dense(cls, name, p):
        """ANY odd p < 2^256 (round 5, fe29.cuh r29_ctx<CURVE_GENERIC>: a curve registered at run time): its nine tight limbs as they are,
        q_k = column * (-p^-1 mod 2^29)."""
        assert p & 1 and 2 < p < 1 << 256, name
        return cls(name, p, [(i, l) for i, l in enumerate(to_limbs(p))], (-pow(p, -1, 1 << W)) % (1 << W))


class AnyPrime(Curve):
    """Every odd p < 2^256 at once, for the INTERVAL execution only: each limb of p is the interval [0, 2^29) (the top one [0, 2^24)), the
    value bound uses p_max = 2^256.  An invariant proven for this object holds for every registered curve: limb and column intervals only grow
    with the limbs of p, and a value bound c p with the products' (c1 p)(c2 p) / 2^261 + p = (c1 c2 p / 2^261 + 1) p grows with p too."""
    def __init__(self):
        self.name, self.p, self.terms, self.qmul, self.tight_sq = "any odd p < 2^256", 1 << 256, None, None, True
        self.term_ranges = [(i, (0, M29 if i < NL - 1 else (1 << 24) - 1)) for i in range(NL)]


def to_limbs(v):
    """Tight limbs of an integer 0 <= v < 2^261 (limbs 0..7 in [0, 2^29), limb 8 the rest)."""
    out = [(v >> (W * i)) & M29 for i in range(NL - 1)]
    out.append(v >> (W * (NL - 1)))
    return out


# p256 = 2^256 - 2^224 + 2^192 + 2^96 - 1: bit 96 = 3*29 + 9, 192 = 6*29 + 18, 224 = 7*29 + 21, 256 = 8*29 + 24
CURVE_P256 = Curve("p256", P256, [(0, -1), (3, 1 << 9), (6, 1 << 18), (7, -(1 << 21)), (8, 1 << 24)], 1)
# secp256k1 = 2^256 - 2^32 - 977: bit 32 = 29 + 3
CURVE_SECP = Curve("secp256k1", SECP, [(0, -977), (1, -8), (8, 1 << 24)], pow(977, -1, 1 << W), tight_sq=False)
CURVE_ANY = AnyPrime()
# registered curves the tests run the exact model on (SEC 2 / RFC 5639 / GB/T 32918 / ANSSI public parameters; all p = 3 mod 4, as the reference's GFp needs)
BRAINPOOL_P256 = 0xA9FB57DBA1EEA9BC3E660A909D838D726E3BF623D52620282013481D1F6E5377
SM2_P = 0xFFFFFFFEFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF00000000FFFFFFFFFFFFFFFF
FRP256_P = 0xF1FD178C0B3AD58F10126DE8CE42435B3961ADBCABC8CA6DE8FCF353D86E9C03


def from_limbs(l):
    return sum(int(x) << (W * i) for i, x in enumerate(l))


# ---------------------------------------------------------------------------------------------------------------- exact execution
class Exact:
    """Concrete execution with the machine's limits asserted; worst_col / worst_limb remember the largest magnitudes met (the witness search's objective)."""
    def __init__(self, curve=CURVE_P256):
        self.cv = curve
        self.worst_col = 0
        self.worst_limb = 0

    def _i32(self, v):
        assert I32[0] <= v <= I32[1], f"limb overflow: {v}"
        if abs(v) > self.worst_limb:
            self.worst_limb = abs(v)
        return v

    def _i64(self, v):
        assert I64[0] <= v <= I64[1], f"column overflow: {v}"
        if abs(v) > self.worst_col:
            self.worst_col = abs(v)
        return v

    def add(self, a, b): return [self._i32(x + y) for x, y in zip(a, b)]
    def sub(self, a, b): return [self._i32(x - y) for x, y in zip(a, b)]
    def dbl(self, a): return [self._i32(x + x) for x in a]

    def norm(self, a, shift=0):
        """One parallel carry pass over (a << shift): limbs 0..7 end in [c_min, 2^29 + c_max), the top limb takes its carry."""
        # the device forms (v << shift) in a wrapping 32-bit register and uses only its low 29 bits; the carry comes from the unshifted limb
        # (v >> (29 - shift)): only the TOP limb's shifted value has to fit
        x = [v << shift for v in a]
        self._i32(x[NL - 1])
        c = [v >> W for v in x]
        out = [x[0] & M29] + [(x[i] & M29) + c[i - 1] for i in range(1, NL - 1)] + [self._i32(x[NL - 1] + c[NL - 2])]
        return out

    def _columns(self, prod):
        """The FIPS column walk shared by mul and sqr: prod(k) -> list of (x, y) factors of column k."""
        cv = self.cv
        q = []
        r = [0] * NL
        acc = 0
        for k in range(2 * NL - 1):
            for x, y in prod(k):
                acc = self._i64(acc + self._i32(x) * self._i32(y))
            for off, c in cv.terms:
                j = k - off
                if off != 0 and 0 <= j < len(q) and j < NL:
                    acc = self._i64(acc + q[j] * c)
            if k < NL:
                qk = (acc * cv.qmul) & M29
                q.append(qk)
                acc = self._i64(acc + qk * dict(cv.terms)[0])
                assert acc & M29 == 0
                acc >>= W
            else:
                r[k - NL] = acc & M29
                acc >>= W
        r[NL - 1] = self._i32(acc)
        return r

    def mul(self, a, b):
        return self._columns(lambda k: [(a[i], b[k - i]) for i in range(max(0, k - NL + 1), min(k, NL - 1) + 1)])

    def sqr(self, a):
        a2 = self.dbl(a)
        def prod(k):
            t = [(a2[i], a[k - i]) for i in range(max(0, k - NL + 1), min(k, NL - 1) + 1) if i < k - i]
            if k % 2 == 0:
                t.append((a[k // 2], a[k // 2]))
            return t
        return self._columns(prod)

    def cswap(self, m, a, b):
        return (b, a) if m else (a, b)

    def vred(self, a):
        """fe29.cuh vred29: v - k p with k = round(top limb / 2^24), through p's sparse signed form."""
        if self.cv.terms is None or len(self.cv.terms) == NL:        # a dense prime (a registered curve): no sparse form, and no need (fe29.cuh vred29)
            return list(a)
        k = (a[NL - 1] + (1 << 23)) >> 24
        out = list(a)
        for off, c in self.cv.terms:
            out[off] = self._i32(out[off] - k * c)
        return out


# ---------------------------------------------------------------------------------------------------------------- interval execution
class Iv:
    """An abstract field element: an interval per limb and an interval for the value."""
    def __init__(self, limbs, val):
        self.l = [tuple(x) for x in limbs]
        self.v = tuple(val)

    def __repr__(self):
        f = lambda x: f"{x / 2**W:+.3f}"
        return "Iv(limbs/2^29: " + " ".join(f"[{f(a)},{f(b)}]" for a, b in self.l) + f"; value/p: [{self.v[0] / P256:+.2f},{self.v[1] / P256:+.2f}])"

    def within(self, other):
        return all(o[0] <= s[0] and s[1] <= o[1] for s, o in zip(self.l, other.l)) and other.v[0] <= self.v[0] and self.v[1] <= other.v[1]


def _iadd(a, b): return (a[0] + b[0], a[1] + b[1])
def _isub(a, b): return (a[0] - b[1], a[1] - b[0])
def _imul(a, b):
    c = (a[0] * b[0], a[0] * b[1], a[1] * b[0], a[1] * b[1])
    return (min(c), max(c))
def _chk(iv, lim, what):
    assert lim[0] <= iv[0] and iv[1] <= lim[1], f"{what} may overflow: [{iv[0]}, {iv[1]}] (2^{max(abs(iv[0]), abs(iv[1])).bit_length()})"
    return iv


class Bounds:
    """Interval execution of the same functions: proves the absence of overflow for EVERY input inside the given intervals."""
    def __init__(self, curve=CURVE_P256):
        self.cv = curve
        self.worst_col = 0
        self.worst_limb = 0
        self.calls = []           # (kind, operand limb boxes, the largest column magnitude of THIS product): what the product-level witnesses aim at

    def _limb(self, iv, what="limb"):
        self.worst_limb = max(self.worst_limb, abs(iv[0]), abs(iv[1]))
        return _chk(iv, I32, what)

    def _col(self, iv, what="column"):
        self.worst_col = max(self.worst_col, abs(iv[0]), abs(iv[1]))
        return _chk(iv, I64, what)

    def _tighten_top(self, x):
        """The top limb is floor((value - low limbs) / 2^232): intersect its interval with what the value interval allows."""
        lo = sum(x.l[i][0] << (W * i) for i in range(NL - 1)); hi = sum(x.l[i][1] << (W * i) for i in range(NL - 1))
        sh = W * (NL - 1)
        t = (-((-(x.v[0] - hi)) // (1 << sh)) if False else (x.v[0] - hi) >> sh, (x.v[1] - lo) >> sh)
        top = (max(x.l[NL - 1][0], t[0]), min(x.l[NL - 1][1], t[1]))
        assert top[0] <= top[1], (x, t)
        x.l[NL - 1] = top
        return x

    def add(self, a, b): return self._tighten_top(Iv([self._limb(_iadd(x, y)) for x, y in zip(a.l, b.l)], _iadd(a.v, b.v)))
    def sub(self, a, b): return self._tighten_top(Iv([self._limb(_isub(x, y)) for x, y in zip(a.l, b.l)], _isub(a.v, b.v)))
    def dbl(self, a): return self._tighten_top(Iv([self._limb((2 * x[0], 2 * x[1])) for x in a.l], (2 * a.v[0], 2 * a.v[1])))

    def norm(self, a, shift=0):
        x = [(v[0] << shift, v[1] << shift) for v in a.l]          # (see Exact.norm: only the top limb's shifted value is materialised)
        self._limb(x[NL - 1], "shifted top limb")
        c = [(v[0] >> W, v[1] >> W) for v in x]
        low = lambda v: (0, M29) if (v[1] - v[0] >= M29 or (v[0] >> W) != (v[1] >> W)) else (v[0] & M29, v[1] & M29)
        out = [low(x[0])] + [self._limb(_iadd(low(x[i]), c[i - 1])) for i in range(1, NL - 1)] + [self._limb(_iadd(x[NL - 1], c[NL - 2]))]
        return self._tighten_top(Iv(out, (a.v[0] << shift, a.v[1] << shift)))

    def _columns(self, prod, tval):
        cv = self.cv
        c0 = dict(cv.term_ranges)[0]
        acc = (0, 0)
        for k in range(2 * NL - 1):
            for x, y in prod(k):
                acc = self._col(_iadd(acc, _imul(x, y)))
            for off, c in cv.term_ranges:
                j = k - off
                if off != 0 and 0 <= j < NL and j < k + 1:
                    acc = self._col(_iadd(acc, _imul((0, M29), c)))
            if k < NL:
                acc = self._col(_iadd(acc, _imul((0, M29), c0)))
            acc = (acc[0] >> W, acc[1] >> W)
        # value: (T + Q p) / R' with 0 <= Q < R'
        R = 1 << RBITS
        val = (tval[0] // R if tval[0] >= 0 else -((-tval[0] + R - 1) // R), (tval[1] + (R - 1) * cv.p) // R + 1)
        out = Iv([(0, M29)] * (NL - 1) + [self._limb(acc, "top limb")], val)
        return self._tighten_top(out)

    def _logged(self, kind, boxes, run):
        before, self.worst_col = self.worst_col, 0
        out = run()
        self.calls.append((kind, boxes, self.worst_col))
        self.worst_col = max(self.worst_col, before)
        return out

    def mul(self, a, b):
        return self._logged("mul", (list(a.l), list(b.l)), lambda: self._columns(
            lambda k: [(a.l[i], b.l[k - i]) for i in range(max(0, k - NL + 1), min(k, NL - 1) + 1)], _imul(a.v, b.v)))

    def sqr(self, a):
        return self._logged("sqr", (list(a.l),), lambda: self._sqr(a))

    def _sqr(self, a):
        a2 = [self._limb((2 * x[0], 2 * x[1]), "doubled limb") for x in a.l]
        def prod(k):
            t = [(a2[i], a.l[k - i]) for i in range(max(0, k - NL + 1), min(k, NL - 1) + 1) if i < k - i]
            if k % 2 == 0:
                s = a.l[k // 2]
                m = max(abs(s[0]), abs(s[1]))
                t.append(((0, m), (0, m)) if s[0] < 0 < s[1] else (s, s))        # a square is never negative
            return t
        m = max(abs(a.v[0]), abs(a.v[1]))
        tv = (0 if a.v[0] < 0 < a.v[1] else min(a.v[0] ** 2, a.v[1] ** 2), m * m)
        return self._columns(prod, tv)

    def vred(self, a):
        """k ranges over what the top limb's interval allows; the new top limb is the rounding remainder, within [-2^23, 2^23]; the value follows
        from the new limbs (the low limbs are whatever they were, moved by k times p's small terms)."""
        if self.cv.terms is None or len(self.cv.terms) == NL:        # a dense prime: the identity (fe29.cuh vred29)
            return a
        assert dict(self.cv.terms)[NL - 1] == 1 << 24
        kr = ((a.l[NL - 1][0] + (1 << 23)) >> 24, (a.l[NL - 1][1] + (1 << 23)) >> 24)
        l = list(a.l)
        for off, c in self.cv.terms:
            l[off] = self._limb((-(1 << 23), 1 << 23)) if off == NL - 1 else self._limb(_isub(l[off], _imul(kr, (c, c))))
        lo = sum(l[i][0] << (W * i) for i in range(NL)); hi = sum(l[i][1] << (W * i) for i in range(NL))
        return Iv(l, (lo, hi))

    def cswap(self, m, a, b):
        j = lambda s, t: (min(s[0], t[0]), max(s[1], t[1]))
        u = Iv([j(s, t) for s, t in zip(a.l, b.l)], j(a.v, b.v))
        return u, Iv(u.l, u.v)


# ---------------------------------------------------------------------------------------------------------------- the ZDAU iteration
def zdau29(E, st, swap):
    """One ladder iteration on the loop state st = dict(x1, x2, dx, y1, dy, z), with the output points exchanged where `swap`.
    Same field values as point.cuh zdau<C> (curve_group.h:120-153): (x1, y1) <- 2 (x1, y1) + (x2, y2), (x2, y2) re-expressed with
    the new z.  Loop-carried besides the coordinates: dx = x1 - x2 and dy = y1 - y2 (differences of tight values; y2 itself is
    y1 - dy and is only materialised after the last iteration)."""
    x1, x2, dx, y1, dy, z = st["x1"], st["x2"], st["dx"], st["y1"], st["dy"], st["z"]
    Cp = E.sqr(dx)
    W1p = E.mul(x1, Cp)
    W2p = E.mul(x2, Cp)
    Dp = E.sqr(dy)
    A1p = E.mul(y1, E.sub(W1p, W2p))
    X3 = E.sub(E.sub(Dp, W1p), W2p)
    u = E.norm(E.sub(X3, W1p))
    Cc = E.sqr(u)
    s = E.sub(dy, u)
    if E.cv.tight_sq:                                         # (secp256k1's sparser reduction leaves room: both squares take their operand as it is)
        s = E.norm(s)
    yp = E.norm(E.sub(E.sub(E.sqr(s), Dp), Cc))               # Y3' + 2 A1'
    A2 = E.dbl(A1p)
    Y3p = E.sub(yp, A2)
    ym = E.norm(E.sub(Y3p, A2))
    C4 = E.norm(Cc, 2)                                        # 4 C, normalised in the same pass
    W1 = E.mul(X3, C4)
    W2 = E.mul(W1p, C4)
    A1 = E.mul(Y3p, E.sub(W1, W2))
    w = E.add(dx, u)
    if E.cv.tight_sq:
        w = E.norm(w)
    zz = E.sub(E.sub(E.sqr(w), Cp), Cc)
    z = E.mul(z, zz)
    ym, yp = E.cswap(swap, ym, yp)
    D = E.sqr(ym)
    Dc = E.sqr(yp)
    W12 = E.add(W1, W2)
    nx1 = E.sub(D, W12)
    nx2 = E.sub(Dc, W12)
    P1 = E.mul(ym, E.sub(W1, nx1))
    P2 = E.mul(yp, E.sub(W1, nx2))
    return {"x1": nx1, "x2": nx2, "dx": E.sub(D, Dc), "y1": E.sub(P1, A1), "dy": E.sub(P1, P2), "z": z}


def ladder_invariant(curve=CURVE_P256):
    """The abstract loop state the ladder maintains (checked by `prove_invariant`): limb intervals in units of 1 and value intervals."""
    p = curve.p
    B = 1 << W
    T8 = 1 << 27                                  # bound on the top limb of everything in the loop (values stay below 2^259)
    lim = lambda lo, hi, vlo, vhi: Iv([(lo * B, hi * B - 1 if hi > 0 else hi * B)] * (NL - 1) + [(-T8, T8)], (vlo * p, vhi * p))
    return {
        "x1": lim(-2, 1, -8, 4), "x2": lim(-2, 1, -8, 4),       # D - (W1 + W2)
        "dx": lim(-1, 1, -4, 4), "dy": lim(-1, 1, -4, 4),       # differences of two tight products
        "y1": lim(-1, 1, -4, 4),                               # P1 - A1
        "z": lim(0, 1, -3, 4),
    }


def prove_invariant(curve=CURVE_P256):
    """One abstract ZDAU from the invariant: nothing overflows and the result lies inside the invariant again.  Returns the
    worst column and limb magnitudes met (as bit lengths) for the record."""
    E = Bounds(curve)
    inv = ladder_invariant(curve)
    out = zdau29(E, {k: Iv(v.l, v.v) for k, v in inv.items()}, True)
    for k in inv:
        assert out[k].within(inv[k]), (k, out[k], inv[k])
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length(), "worst_column": E.worst_col, "out": out}


def prove_pow_chain(curve=CURVE_P256):
    """gcurve.cuh gc_pow29 / point.cuh fe_sqrt_candidate29: a^e as a chain of sqr29 / mul29 whose every factor is the output of a product (or of enter29, a
    product too).  Closure: with T = what a product of canonical-width operands can return, a square or a product of members of T lies in T again, nothing
    overflows on the way, and a member of T is in leave29's domain.  (The ladder's invariant is wider than T -- differences of products -- which is why this
    chain needs no carry pass at all.)"""
    E = Bounds(curve)
    p = curve.p
    B = 1 << W
    canonical = Iv([(0, B - 1)] * (NL - 1) + [(0, (1 << 24) - 1)], (0, p - 1))              # to29 of a canonical residue; the constants 2^266 mod p, 2^256 mod p too
    T = Iv([(0, B - 1)] * (NL - 1) + [(-1, (1 << 24) + (1 << 20))], (-(p >> 5), p + (p >> 4)))   # what a product returns: tight limbs, a value in (-p/32, 17 p/16)
    for v in (E.mul(canonical, canonical), E.sqr(T), E.mul(T, T), E.mul(T, canonical)):     # enter29; a square, a product of two members; one by a table power
        assert v.within(T), (v, T)
    out = E.mul(T, canonical)                                                                # leave29's product by 2^256 mod p ...
    assert -p < out.v[0] and out.v[1] < 2 * p, out                                           # ... lands in canon29's domain
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length(), "factor": T}


# ---------------------------------------------------------------------------------------------------------------- the combs' mixed addition
def madd29(E, X1, Y1, Z1, x2, y2):
    """fe29.cuh madd29<C> (= madd29_hr + madd29_finish): Jacobian (X1, Y1, Z1) + affine (x2, y2), Hankerson-Menezes-Vanstone Alg. 3.22 as
    point.cuh madd_hmv (8M + 3S); statement for statement the device function."""
    Z1Z1 = E.sqr(Z1)
    U2 = E.mul(x2, Z1Z1)
    S2 = E.mul(y2, E.mul(Z1Z1, Z1))
    H = E.norm(E.sub(U2, X1))
    r = E.norm(E.sub(S2, Y1))
    HH = E.sqr(H)
    HHH = E.mul(H, HH)
    V = E.mul(X1, HH)
    Z3 = E.mul(Z1, H)
    X3 = E.sub(E.sub(E.sqr(r), HHH), E.dbl(V))
    Y3 = E.sub(E.mul(r, E.norm(E.sub(V, X3))), E.mul(Y1, HHH))
    return X3, Y3, Z3


def comb_invariant(curve=CURVE_P256):
    """The accumulator of the fixed-base combs between two additions, and what a table read hands madd29: X = r^2 - HHH - 2V, Y a difference of
    two products, Z a product; a table coordinate is to29 of a canonical residue (tight, value in [0, p)), its y possibly negated."""
    p = curve.p
    B = 1 << W
    T8 = 1 << 27
    lim = lambda lo, hi, vlo, vhi: Iv([(lo * B, hi * B)] * (NL - 1) + [(-T8, T8)], (vlo * p, vhi * p))
    return {"X": lim(-3, 1, -8, 5), "Y": lim(-1, 1, -4, 4), "Z": lim(0, 1, -3, 4), "tx": lim(0, 1, 0, 1), "ty": lim(-1, 1, -1, 1)}


def prove_comb_invariant(curve=CURVE_P256):
    """One abstract madd29 from the invariant: no limb or column overflows, the sum lies inside the invariant again.  (The first entry -- tight
    x, +-y, Z = the constant 2^261 mod p -- lies inside it trivially.)  All three carry passes are needed: without any one of them the proof fails."""
    E = Bounds(curve)
    inv = comb_invariant(curve)
    c = lambda k: Iv(inv[k].l, inv[k].v)
    X3, Y3, Z3 = madd29(E, c("X"), c("Y"), c("Z"), c("tx"), c("ty"))
    assert X3.within(inv["X"]) and Y3.within(inv["Y"]) and Z3.within(inv["Z"]), (X3, Y3, Z3)
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length()}


def madd_field(p, X1, Y1, Z1, x2, y2):
    """The same formulas on integers mod p (Jacobian x = X / Z^2, y = Y / Z^3)."""
    Z1Z1 = Z1 * Z1 % p; U2 = x2 * Z1Z1 % p; S2 = y2 * Z1Z1 * Z1 % p
    H = (U2 - X1) % p; r = (S2 - Y1) % p
    HH = H * H % p; HHH = H * HH % p; V = X1 * HH % p
    X3 = (r * r - HHH - 2 * V) % p
    return X3, (r * (V - X3) - Y1 * HHH) % p, Z1 * H % p


# ---------------------------------------------------------------------------------------------------------------- the variable-base window loop
def jdbl29(E, X, Y, Z):
    """fe29.cuh jdbl29<C> (a = -3 for P-256, a = 0 for secp256k1), statement for statement."""
    Yn = E.norm(Y)
    YY = E.sqr(Yn)
    G = E.norm(YY, 2)
    B = E.mul(X, G)
    if E.cv is CURVE_P256:
        delta = E.sqr(Z)
        t = E.mul(E.sub(X, delta), E.norm(E.add(X, delta)))
    else:
        t = E.sqr(E.norm(X))
    alpha = E.norm(E.add(E.dbl(t), t))
    Z3 = E.mul(E.dbl(Yn), Z)
    X3 = E.sub(E.sqr(alpha), E.dbl(B))
    E8 = E.dbl(E.norm(E.sqr(YY), 2))
    Y3 = E.vred(E.sub(E.mul(alpha, E.sub(B, X3)), E8))
    return E.vred(X3), Y3, Z3


def dbl_add29(E, X, Y, Z, x2, y2):
    """fe29.cuh dbl_add29<C>: 2 (X, Y, Z) + (x2, y2) as (R + T) + R, the second addition co-Z."""
    Z1Z1 = E.sqr(Z); U2 = E.mul(x2, Z1Z1); S2 = E.mul(y2, E.mul(Z1Z1, Z))
    H = E.norm(E.sub(U2, X)); r = E.norm(E.sub(S2, Y))
    HH = E.sqr(H); HHH = E.mul(H, HH); V = E.mul(X, HH); Yh = E.mul(Y, HHH)
    X3 = E.sub(E.sub(E.sqr(r), HHH), E.dbl(V))
    Y3 = E.sub(E.mul(r, E.norm(E.sub(V, X3))), Yh)
    Z3 = E.mul(Z, H)
    dx = E.norm(E.sub(X3, V)); dy = E.norm(E.sub(Y3, Yh))
    Cc = E.sqr(dx); W1 = E.mul(X3, Cc); W2 = E.mul(V, Cc)
    A1 = E.mul(Y3, E.sub(W1, W2))
    Qx = E.sub(E.sub(E.sqr(dy), W1), W2)
    Qy = E.vred(E.sub(E.mul(dy, E.sub(W1, Qx)), A1))
    return E.vred(Qx), Qy, E.mul(Z3, dx)


def window_invariant(curve=CURVE_P256):
    """The accumulator of k_varwin_mult_odd between two point operations: X and Y fresh from vred29 (|value| <= 0.6 p; limbs of a difference of
    products, moved by at most a few 2^21 by the reduction), Z a product."""
    p = curve.p
    B = 1 << W
    lim = lambda lo, hi, top, vlo, vhi: Iv([(int(lo * B), int(hi * B))] * (NL - 1) + [(-top, top)], (int(vlo * p), int(vhi * p)))
    return {"X": lim(-2.25, 1.25, 1 << 25, -1.05, 1.05), "Y": lim(-2.25, 1.25, 1 << 25, -1.05, 1.05), "Z": lim(0, 1, 1 << 27, -3, 4),
            "tx": lim(0, 1, 1 << 27, 0, 1), "ty": lim(-1, 1, 1 << 27, -1, 1)}


def prove_window_invariant(curve=CURVE_P256):
    """One window of the loop -- three jdbl29 and one dbl_add29 -- on intervals: no overflow, and after EVERY one of the four operations the
    accumulator lies inside the invariant again (so any mix of doublings and additions is covered, the table-entry start included)."""
    E = Bounds(curve)
    inv = window_invariant(curve)
    c = lambda k: Iv(inv[k].l, inv[k].v)
    inside = lambda X, Y, Z: X.within(inv["X"]) and Y.within(inv["Y"]) and Z.within(inv["Z"])
    X, Y, Z = c("X"), c("Y"), c("Z")
    for _ in range(3):
        X, Y, Z = jdbl29(E, X, Y, Z)
        assert inside(X, Y, Z), (X, Y, Z)
    X, Y, Z = dbl_add29(E, c("X"), c("Y"), c("Z"), c("tx"), c("ty"))
    assert inside(X, Y, Z), (X, Y, Z)
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length()}


# ---------------------------------------------------------------------------------------------------------------- the default GLV loop (secp256k1)
def madd29v(E, X1, Y1, Z1, x2, y2, with_hr=False):
    """fe29.cuh madd29_hr<C> + madd29v_finish<C>: madd29, then one more carry pass on X3 and the value reduction on X3 and Y3."""
    Z1Z1 = E.sqr(Z1); U2 = E.mul(x2, Z1Z1); S2 = E.mul(y2, E.mul(Z1Z1, Z1))
    H = E.norm(E.sub(U2, X1)); r = E.norm(E.sub(S2, Y1))
    HH = E.sqr(H); HHH = E.mul(H, HH); V = E.mul(X1, HH)
    Z3 = E.mul(Z1, H)
    X3 = E.sub(E.sub(E.sqr(r), HHH), E.dbl(V))
    Y3 = E.sub(E.mul(r, E.norm(E.sub(V, X3))), E.mul(Y1, HHH))
    out = (E.vred(E.norm(X3)), E.vred(Y3), Z3)
    return out + (H, r) if with_hr else out


def is_zero29(E, v):
    """fe29.cuh is_zero29<C> on the Exact machine: vred29, a sequential carry pass over limbs 0..7, all nine limbs zero."""
    v = E.vred(v)
    for i in range(NL - 1):
        v[i + 1] = E._i32(v[i + 1] + (v[i] >> W)); v[i] &= M29
    return not any(v)


def prove_glv_invariant(curve=CURVE_SECP):
    """k_varwin_mult_glv on 29-bit limbs: jdbl29 and madd29v in any order (four doublings, two additions per window; the accumulator may also be
    replaced by a table entry with Z = 2^261 mod p, or by jdbl29 of one) keep the window loop's invariant; x2 may be a table coordinate or its
    product with beta (value up to 1.05 p).  And the operands of is_zero29, H and r, are small enough for its argument: after vred29 |v| < p."""
    E = Bounds(curve)
    inv = window_invariant(curve)
    p = curve.p
    c = lambda k: Iv(inv[k].l, inv[k].v)
    inside = lambda P: all(P[i].within(inv[k]) for i, k in enumerate("XYZ"))
    tx = Iv(inv["tx"].l, (0, int(1.05 * p)))
    assert inside(jdbl29(E, c("X"), c("Y"), c("Z")))
    X3, Y3, Z3, H, r = madd29v(E, c("X"), c("Y"), c("Z"), tx, c("ty"), with_hr=True)
    assert inside((X3, Y3, Z3)), (X3, Y3, Z3)
    assert inside(jdbl29(E, tx, c("ty"), Iv(inv["Z"].l, (0, p))))                           # the tangent at a table point (R = T)
    for v in (H, r):
        w = E.vred(v)
        assert -p < w.v[0] and w.v[1] < p, w
        t = w.l[NL - 1]                                                                      # the sequential pass: the carries stay small
        assert all(abs(b) < (1 << 31) - (1 << 3) for iv in w.l for b in iv) and abs(t[0]) < 1 << 30 and abs(t[1]) < 1 << 30
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length()}


def zaddu29(E, x1, y1, x2, y2, z):
    """fe29.cuh zaddu29<C>: the co-Z addition with update (curve_group.h:91-116 ZADDU, 5M + 2S) -- (x1, y1) + (x2, y2) over the common z; returns the
    sum, (x1, y1) re-expressed over the new z = z dx, and dx (the ratio of the two Z)."""
    dx = E.norm(E.sub(x1, x2))
    Cc = E.sqr(dx)
    W1 = E.mul(x1, Cc)
    W2 = E.mul(x2, Cc)
    dy = E.norm(E.sub(y1, y2))
    D = E.sqr(dy)
    A1 = E.mul(y1, E.sub(W1, W2))
    rx = E.sub(E.sub(D, W1), W2)
    ry = E.sub(E.mul(dy, E.sub(W1, rx)), A1)
    return rx, ry, W1, A1, E.mul(z, dx), dx


def iso_chain_invariant(curve=CURVE_SECP):
    """k_varwin_table_iso's forward chain: the running P is a pair of products (tight), the last multiple is jdbl29's output or a zaddu29 sum."""
    p = curve.p
    B = 1 << W
    lim = lambda lo, hi, top, vlo, vhi: Iv([(int(lo * B), int(hi * B))] * (NL - 1) + [(-top, top)], (int(vlo * p), int(vhi * p)))
    return {"px": lim(0, 1, 1 << 27, -0.25, 1.25), "py": lim(0, 1, 1 << 27, -0.25, 1.25), "z": lim(0, 1, 1 << 27, -0.25, 1.25),
            "mx": lim(-2.25, 1.25, 1 << 27, -2.5, 1.5), "my": lim(-2.25, 1.25, 1 << 27, -1.5, 1.5)}


def prove_iso_table(curve=CURVE_SECP):
    """k_varwin.inc k_varwin_table_iso on intervals.  Forward: 2P = jdbl29(P), P re-expressed over its Z (three products), then zaddu29 six times -- the
    chain's invariant (iso_chain_invariant) holds from the start and maps into itself.  Backward: everything handed to canon29 has a value in its
    domain (-p, 2p), and the walk's products (tight f, f^2, f^3, a carry-passed dx, the lazy X_k, Y_k of the chain) stay inside the machine."""
    E = Bounds(curve)
    winv = window_invariant(curve)
    inv = iso_chain_invariant(curve)
    p = curve.p
    c = lambda k: Iv(inv[k].l, inv[k].v)
    ok = lambda v: -p < v.v[0] and v.v[1] < 2 * p                     # + p, then two conditional subtractions of p: [0, p) for anything in (-p, 2p)
    tx, ty = Iv(winv["tx"].l, winv["tx"].v), Iv(winv["ty"].l, (0, p))
    X2, Y2, Z2 = jdbl29(E, tx, ty, Iv(winv["Z"].l, (0, p)))                                # 2P from the tight input point, Z = 1
    ZZ = E.sqr(Z2)
    px, py = E.mul(tx, ZZ), E.mul(ty, E.mul(ZZ, Z2))                                       # P over Z_2
    assert X2.within(inv["mx"]) and Y2.within(inv["my"]) and Z2.within(inv["z"]) and px.within(inv["px"]) and py.within(inv["py"])
    rx, ry, W1, A1, z, dx = zaddu29(E, c("px"), c("py"), c("mx"), c("my"), c("z"))
    assert rx.within(inv["mx"]) and ry.within(inv["my"]) and W1.within(inv["px"]) and A1.within(inv["py"]) and z.within(inv["z"])
    assert not ok(rx) and ok(E.vred(rx)) and ok(E.vred(ry)) and ok(z)   # 8P goes to canon29: its X = D - W1 - W2 reaches below -p, so vred29 first
    tight = lambda: Iv(winv["tx"].l, (-p // 4, 5 * p // 4))           # a product (of a factor that may be negative, or as wide as dx)
    f = E.mul(tight(), dx)                                            # f_k = f_(k+1) dx_k
    f2 = E.sqr(tight())
    f3 = E.mul(tight(), tight())
    for v in (f, f2, f3, E.mul(c("mx"), tight()), E.mul(c("my"), tight()), E.mul(tx, tight())):      # kP waits in scratch as it was born: lazy limbs times f^2, f^3
        assert ok(v) and v.within(tight()), v
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length()}


def jdbl_field(p, a, X, Y, Z):
    YY = Y * Y % p; B = 4 * X * YY % p
    alpha = (3 * X * X + a * pow(Z, 4, p)) % p
    X3 = (alpha * alpha - 2 * B) % p
    return X3, (alpha * (B - X3) - 8 * YY * YY) % p, 2 * Y * Z % p


def dbl_add_field(p, X, Y, Z, x2, y2):
    X3, Y3, Z3 = madd_field(p, X, Y, Z, x2, y2)                   # R + T
    H = (x2 * Z * Z - X) % p
    V, Yh = X * H * H % p, Y * pow(H, 3, p) % p                     # R over Z3
    dx, dy = (X3 - V) % p, (Y3 - Yh) % p
    Cc = dx * dx % p; W1 = X3 * Cc % p; W2 = V * Cc % p
    Qx = (dy * dy - W1 - W2) % p
    return Qx, (dy * (W1 - Qx) - Y3 * (W1 - W2)) % p, Z3 * dx % p


# ---------------------------------------------------------------------------------------------------------------- the window loop of a registered curve (round 5)
def gjdbl29(E, X, Y, Z, Wc, wout=True):
    """fe29.cuh gjdbl29<C, WOUT>: the doubling for ANY a in modified Jacobian coordinates, Wc = a Z^4 beside the point; statement for statement.
    8 Y^4 is 2 (2 YY)^2 -- a square of a carry-passed double, never 8 x a product."""
    Yn = E.norm(Y)
    YY = E.sqr(Yn)
    G = E.norm(YY, 2)
    B = E.mul(X, G)
    XX = E.sqr(E.norm(X))
    alpha = E.norm(E.add(E.add(E.dbl(XX), XX), Wc))
    Z3 = E.mul(E.dbl(Yn), Z)
    X3 = E.sub(E.sqr(alpha), E.dbl(B))
    E4 = E.sqr(E.norm(YY, 1))
    Y3 = E.sub(E.mul(alpha, E.norm(E.sub(B, X3))), E.dbl(E4))
    W3 = E.mul(E.norm(E4, 2), Wc) if wout else Wc
    return X3, Y3, Z3, W3


def gwindow_invariant(curve=CURVE_ANY):
    """The accumulator of k_gvarwin.hip k_gvw_mult between two point operations, in units of p_max = 2^256 for CURVE_ANY: X = alpha^2 - 2B or a co-Z sum,
    Y a difference of products, Z and W = a' Z^4 products; a table coordinate is to29 of a canonical residue, its y possibly negated; ap = a Zg^4 a product."""
    p = curve.p
    B = 1 << W
    lim = lambda lo, hi, top, vlo, vhi: Iv([(int(lo * B), int(hi * B))] * (NL - 1) + [(-top, top)], (int(vlo * p), int(vhi * p)))
    return {"X": lim(-2, 1, 1 << 26, -3.25, 3.25), "Y": lim(-2, 1, 1 << 26, -3.3, 2.0), "Z": lim(0, 1, 1 << 25, -0.55, 1.55), "W": lim(0, 1, 1 << 25, -0.2, 1.35),
            "tx": lim(0, 1, 1 << 24, 0, 1), "ty": lim(-1, 1, 1 << 24, -1, 1), "ap": lim(0, 1, 1 << 25, -0.1, 1.1)}


def gw_of_z(E, Z, ap):
    """k_gvarwin.hip: W = a' Z^4 after an addition (the doublings that follow carry it along)."""
    return E.mul(ap, E.sqr(E.sqr(Z)))


def prove_gwindow_invariant(curve=CURVE_ANY):
    """One window of k_gvw_mult on intervals, for every odd p < 2^256 at once: W from Z, gjdbl29 with W carried (twice), gjdbl29 without, dbl_add29 (its
    value reductions are the identity on a dense prime).  No limb or column overflows, and after EVERY operation the accumulator -- W included -- lies
    inside the invariant again, so the start from a table entry (tight x, +-y, Z = 2^261 mod p, W = a') is covered too."""
    E = Bounds(curve)
    inv = gwindow_invariant(curve)
    c = lambda k: Iv(inv[k].l, inv[k].v)
    inside = lambda X, Y, Z: X.within(inv["X"]) and Y.within(inv["Y"]) and Z.within(inv["Z"])
    Wc = gw_of_z(E, c("Z"), c("ap"))
    assert Wc.within(inv["W"]), Wc
    assert c("ap").within(inv["W"])                                          # the start: W = a' beside Z = 1
    X, Y, Z, Wc = c("X"), c("Y"), c("Z"), c("W")
    for wout in (True, True, False):
        X, Y, Z, Wn = gjdbl29(E, X, Y, Z, Wc, wout)
        assert inside(X, Y, Z), (X, Y, Z)
        if wout:
            assert Wn.within(inv["W"]), Wn
            Wc = Wn
        X, Y, Z, Wc = c("X"), c("Y"), c("Z"), c("W")                         # every doubling from the whole invariant, not from the one before
    X, Y, Z = dbl_add29(E, c("X"), c("Y"), c("Z"), c("tx"), c("ty"))
    assert inside(X, Y, Z), (X, Y, Z)
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length()}


def prove_gtable(curve=CURVE_ANY):
    """k_gvarwin.hip k_gvw_table on intervals: 2P = gjdbl29(P) from the tight input point (Z = 1, W = a), P over Z_2 by three products, then seven zaddu29
    -- the first with the lazy 2P as its (x1, y1), the others with the re-expressed (tight) one -- and the walk back: f, f^2, f^3, X_k f^2, Y_k f^3.
    Nothing overflows, and whatever is handed to canon29 lies in its domain (-p, 2p) for every p the windowed route is registered for
    (n >= 2^255, so p > 2^255 - 2^129 by Hasse): a product's value is T / 2^261 + [0, p), so T / 2^261 in (-p_min, p_min) is what to show."""
    E = Bounds(curve)
    inv = gwindow_invariant(curve)
    p = curve.p
    pmin = (1 << 255) - (1 << 129)
    c = lambda k: Iv(inv[k].l, inv[k].v)
    ok = lambda v: -pmin < v.v[0] and v.v[1] - p < pmin                   # v.v[1] = sup(T / 2^261) + p_max
    one = Iv(inv["tx"].l, inv["tx"].v)
    ty = Iv(inv["tx"].l, inv["tx"].v)
    X2, Y2, Z2, _ = gjdbl29(E, c("tx"), ty, one, c("ap"), False)             # (a 2^261 mod p itself is enter29's output: inside "ap")
    ZZ = E.sqr(Z2)
    px, py = E.mul(c("tx"), ZZ), E.mul(ty, E.mul(ZZ, Z2))
    lazy = []
    x1, y1, x2, y2, z = X2, Y2, px, py, Z2
    for _ in range(7):
        lazy.append((x2, y2))
        rx, ry, W1, A1, z, dx = zaddu29(E, x1, y1, x2, y2, z)
        x1, y1, x2, y2 = W1, A1, rx, ry
        assert z.within(inv["Z"]), z
    tight = lambda: Iv(inv["tx"].l[:NL - 1] + [(-(1 << 22), (1 << 24) + (1 << 22))], (-p // 4, 5 * p // 4))   # any product met here
    for v in (z, dx, W1, A1):
        assert v.within(tight()) or v is dx, v
    f = E.mul(tight(), dx)
    f2 = E.sqr(tight())
    f3 = E.mul(tight(), tight())
    for v in (f, f2, f3):
        assert v.within(tight()), v
    for X, Y in lazy + [(x2, y2)]:
        for v in (E.mul(X, tight()), E.mul(Y, tight())):                     # the last multiple: times the field's 1, a tight constant
            assert ok(v) and v.within(tight()), v
    assert ok(z)                                                             # Zg goes to canon29 as it is: a product
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length()}


def gtable_invariant(curve=CURVE_ANY):
    """What k_gvw_table hands zaddu29 (the witness search's box): (x1, y1) the doubled point as gjdbl29 leaves it or its re-expressed (tight) form, (x2, y2)
    the running odd multiple -- products at first, co-Z sums afterwards -- and z a product."""
    p = curve.p
    box = lambda lo, hi, top, vlo, vhi: Iv([(lo, hi)] * (NL - 1) + [(-top, top)], (int(vlo * p), int(vhi * p)))
    return {"x1": box(-2 * M29, M29, 1 << 26, -3.25, 3.25), "y1": box(-2 * M29, M29, 1 << 26, -3.3, 2.0), "x2": box(-2 * M29, M29, 1 << 26, -2.5, 1.6),
            "y2": box(-M29, M29, 1 << 26, -1.5, 1.5), "z": box(0, M29, 1 << 25, -0.55, 1.55)}


def prove_gtable_box(curve=CURVE_ANY):
    """zaddu29 from the whole of gtable_invariant (wider than any state k_gvw_table meets: both points lazy at once): no overflow."""
    E = Bounds(curve)
    inv = gtable_invariant(curve)
    zaddu29(E, *(Iv(inv[k].l, inv[k].v) for k in ("x1", "y1", "x2", "y2", "z")))
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length()}


def gjdbl_field(p, X, Y, Z, Wc):
    """Modified Jacobian doubling on integers mod p: (X3, Y3, Z3, W3) with W = a Z^4."""
    YY = Y * Y % p; B = 4 * X * YY % p; E8 = 8 * YY * YY % p
    alpha = (3 * X * X + Wc) % p
    X3 = (alpha * alpha - 2 * B) % p
    return X3, (alpha * (B - X3) - E8) % p, 2 * Y * Z % p, 2 * E8 * Wc % p


# ---------------------------------------------------------------------------------------------------------------- the complete addition law (a = 0, b = 7)
def mul21_29(E, x):
    a = E.norm(x, 2)
    b = E.norm(a, 2)
    return E.vred(E.add(E.add(a, b), x))


def pdbl29(E, X, Y, Z):
    """fe29.cuh pdbl29<C>: homogeneous projective doubling of Renes-Costello-Batina for a = 0, 3b = 21 (6M + 2S)."""
    Yn = E.norm(Y); Zn = E.norm(Z)
    yy = E.sqr(Yn); zz = E.sqr(Zn)
    xy = E.mul(X, Yn); yz = E.mul(Yn, Zn)
    t = E.norm(mul21_29(E, zz))
    m = E.sub(yy, E.add(E.dbl(t), t))
    q = E.norm(E.add(yy, t))
    X3 = E.vred(E.dbl(E.mul(xy, m)))
    Y3 = E.vred(E.add(E.mul(m, q), E.dbl(E.norm(E.mul(yy, t), 2))))
    Z3 = E.vred(E.dbl(E.norm(E.mul(yy, yz), 2)))
    return X3, Y3, Z3


def padd29(E, X, Y, Z, x2, y2):
    """fe29.cuh padd29<C>: (X : Y : Z) + affine (x2, y2), the complete mixed addition (11M)."""
    Xn = E.norm(X); Yn = E.norm(Y); Zn = E.norm(Z)
    t0 = E.mul(Xn, x2); t1 = E.mul(Yn, y2)
    t3 = E.sub(E.sub(E.mul(E.add(Xn, Yn), E.norm(E.add(x2, y2))), t0), t1)
    t4 = E.norm(E.add(E.mul(y2, Zn), Yn))
    t5 = E.norm(E.add(E.mul(x2, Zn), Xn))
    z3b = E.norm(mul21_29(E, Zn))
    A = E.sub(t1, z3b); B = E.add(t1, z3b)
    Cc = mul21_29(E, t5)
    t03 = E.norm(E.add(E.dbl(t0), t0))
    X3 = E.vred(E.sub(E.mul(t3, A), E.mul(Cc, t4)))
    Y3 = E.vred(E.add(E.mul(t03, Cc), E.mul(B, A)))
    Z3 = E.vred(E.add(E.mul(t4, B), E.mul(t03, t3)))
    return X3, Y3, Z3


def complete_invariant(curve=CURVE_SECP):
    p = curve.p
    B = 1 << W
    lim = lambda lo, hi, top, vlo, vhi: Iv([(int(lo * B), int(hi * B))] * (NL - 1) + [(-top, top)], (int(vlo * p), int(vhi * p)))
    inv = {k: lim(-2.25, 3.25, 1 << 25, -1.05, 1.05) for k in "XYZ"}
    inv.update({"tx": lim(0, 1, 1 << 27, 0, 1), "ty": lim(-1, 1, 1 << 27, -1, 1)})
    return inv


def prove_complete_invariant(curve=CURVE_SECP):
    """pdbl29 and padd29 from the invariant (the neutral element (0 : 1 : 0) in tight limbs lies inside it): no overflow, closed."""
    E = Bounds(curve)
    inv = complete_invariant(curve)
    c = lambda k: Iv(inv[k].l, inv[k].v)
    inside = lambda P: all(P[i].within(inv[k]) for i, k in enumerate("XYZ"))
    assert inside(pdbl29(E, c("X"), c("Y"), c("Z")))
    assert inside(padd29(E, c("X"), c("Y"), c("Z"), c("tx"), c("ty")))
    return {"worst_column_bits": E.worst_col.bit_length(), "worst_limb_bits": E.worst_limb.bit_length()}


def pdbl_field(p, X, Y, Z):
    yy = Y * Y % p; t = 21 * Z * Z % p; m = (yy - 3 * t) % p; q = (yy + t) % p
    return 2 * X * Y * m % p, (m * q + 8 * yy * t) % p, 8 * yy * Y * Z % p


def padd_field(p, X, Y, Z, x2, y2):
    t0 = X * x2 % p; t1 = Y * y2 % p; t3 = ((X + Y) * (x2 + y2) - t0 - t1) % p
    t4 = (y2 * Z + Y) % p; t5 = (x2 * Z + X) % p
    A = (t1 - 21 * Z) % p; B = (t1 + 21 * Z) % p; Cc = 21 * t5 % p
    return (t3 * A - Cc * t4) % p, (3 * t0 * Cc + B * A) % p, (t4 * B + 3 * t0 * t3) % p


# ---------------------------------------------------------------------------------------------------------------- witnesses (round 5)
# The interval proofs bound every column by 2^63 over the whole invariant box; nothing so far fed the DEVICE an input anywhere near those extremes
# (random and digit-pattern operands enter the loops as tight limbs).  OPS names, for every function a proof covers, its inputs, the invariant
# they live in and the model function; witness_search climbs over the box's vertices (every limb at an end of its interval, the top limb chosen so
# that the value stays inside its interval) towards the largest column the EXACT model meets.  tests/golden/fe29_witnesses.json holds the
# states found (tools/make_witnesses.py), their exact outputs and how close they come to the proven bound; ecsimd_hip_fe29_raw runs them on
# the device, limb for limb (tests/test_gpu_witness.py).
def _op_zdau(E, a, sw): o = zdau29(E, dict(zip(("x1", "x2", "dx", "y1", "dy", "z"), a)), sw); return [o[k] for k in ("x1", "x2", "dx", "y1", "dy", "z")]
def _op3(fn): return lambda E, a, sw: list(fn(E, *a))
OPS = {   # name: (op code of ecsimd_hip_fe29_raw, input names, invariant, model function, outputs)
    "zdau": (0, ("x1", "x2", "dx", "y1", "dy", "z"), ladder_invariant, _op_zdau, 6),
    "madd": (1, ("X", "Y", "Z", "tx", "ty"), comb_invariant, _op3(madd29), 3),
    "jdbl": (2, ("X", "Y", "Z"), window_invariant, _op3(jdbl29), 3),
    "dbl_add": (3, ("X", "Y", "Z", "tx", "ty"), window_invariant, _op3(dbl_add29), 3),
    "maddv": (4, ("X", "Y", "Z", "tx", "ty"), window_invariant, _op3(lambda E, *a: madd29v(E, *a)), 3),
    "pdbl": (5, ("X", "Y", "Z"), None, _op3(pdbl29), 3),
    "padd": (6, ("X", "Y", "Z", "tx", "ty"), None, _op3(padd29), 3),
}
_dense = lambda cv: cv.terms is None or len(cv.terms) == NL
_iso_as_zaddu = lambda cv: (lambda i: {"x1": i["px"], "y1": i["py"], "x2": i["mx"], "y2": i["my"], "z": i["z"]})(iso_chain_invariant(cv))
OPS["dbl_add"] = OPS["dbl_add"][:2] + (lambda cv: gwindow_invariant(cv) if _dense(cv) else window_invariant(cv),) + OPS["dbl_add"][3:]
OPS["gjdbl"] = (9, ("X", "Y", "Z", "W"), gwindow_invariant, lambda E, a, sw: list(gjdbl29(E, *a, wout=bool(sw))), 4)          # swap = WOUT
OPS["zaddu"] = (10, ("x1", "y1", "x2", "y2", "z"), lambda cv: gtable_invariant(cv) if _dense(cv) else _iso_as_zaddu(cv), _op3(zaddu29), 6)
OPS["pdbl"] = OPS["pdbl"][:2] + (lambda cv: complete_invariant(cv),) + OPS["pdbl"][3:]
OPS["padd"] = OPS["padd"][:2] + (lambda cv: complete_invariant(cv),) + OPS["padd"][3:]


def _vertex(iv, rng, choice=None):
    """A concrete state inside the abstract element iv: limbs 0..7 at an end of their interval (choice[i] picks which; random where None), the top
    limb at the end of what the VALUE interval leaves of its own interval (towards choice[8])."""
    ch = [rng.getrandbits(1) for _ in range(NL)] if choice is None else list(choice)
    low = [iv.l[i][ch[i]] for i in range(NL - 1)]
    S = sum(v << (W * i) for i, v in enumerate(low))
    sh = W * (NL - 1)
    lo = max(iv.l[NL - 1][0], -((-(iv.v[0] - S)) // (1 << sh)))
    hi = min(iv.l[NL - 1][1], (iv.v[1] - S) >> sh)
    if lo > hi:
        return None, ch
    return low + [hi if ch[NL - 1] else lo], ch


def product_witnesses(curve, prove, top=3, seed=1, steps=400):
    """The products and squares of one interval proof whose columns come closest to 2^63, each with a concrete operand pair -- limbs at the ends of the
    boxes the proof hands that very call -- found by climbing over the box's vertices on the exact model.  [(kind, a, b or None, exact result, achieved
    worst column, proven worst column of that call)]."""
    import random
    rng = random.Random(seed)
    E0 = prove(curve)
    calls = E0 if isinstance(E0, list) else None
    assert calls is not None
    out = []
    for kind, boxes, proven in sorted(calls, key=lambda c: -c[2])[:top]:
        def run(ch):
            ops = [[box[i][ch[j][i]] for i in range(NL)] for j, box in enumerate(boxes)]
            E = Exact(curve)
            r = E.mul(ops[0], ops[1]) if kind == "mul" else E.sqr(ops[0])
            return ops, r, E.worst_col
        # start where every limb has its largest magnitude (a sum of nine products is largest when the factors are), then climb
        ch = [[int(abs(b[1]) >= abs(b[0])) for b in box] for box in boxes]
        best = run(ch)
        for _ in range(steps):
            j, i = rng.randrange(len(boxes)), rng.randrange(NL)
            t = [list(x) for x in ch]
            t[j][i] ^= 1
            r = run(t)
            if r[2] >= best[2]:
                best, ch = r, t
        out.append((kind, best[0][0], best[0][1] if kind == "mul" else None, best[1], best[2], proven))
    return out


def proof_calls(prove_fn):
    """Adapter: run an interval proof and hand back the log of its products (Bounds.calls)."""
    def run(curve):
        holder = {}
        orig = Bounds.__init__
        def init(self, cv=CURVE_P256):
            orig(self, cv); holder["E"] = self
        Bounds.__init__ = init
        try:
            prove_fn(curve)
        finally:
            Bounds.__init__ = orig
        return holder["E"].calls
    return run


def witness_search(op, curve, seed, steps=1500, swap=False):
    """Hill-climb over vertices of the invariant box of `op` for the largest column of the exact model.  Returns (inputs, outputs, worst column)."""
    import random
    rng = random.Random(seed)
    code, names, inv_fn, fn, nout = OPS[op]
    inv = inv_fn(curve)
    ivs = [inv[n] for n in names]

    def run(choices):
        st = []
        for iv, ch in zip(ivs, choices):
            v, _ = _vertex(iv, rng, ch)
            if v is None:
                return None
            st.append(v)
        E = Exact(curve)
        out = fn(E, [list(x) for x in st], swap)
        return st, out, E.worst_col

    best = None
    while best is None:
        choices = [[rng.getrandbits(1) for _ in range(NL)] for _ in ivs]
        best = run(choices)
    for _ in range(steps):
        c, i = rng.randrange(len(ivs)), rng.randrange(NL)
        trial = [list(x) for x in choices]
        trial[c][i] ^= 1
        if rng.random() < 0.3:                                     # now and then two flips at once: columns couple neighbouring limbs
            c2, i2 = rng.randrange(len(ivs)), rng.randrange(NL)
            trial[c2][i2] ^= 1
        r = run(trial)
        if r is not None and r[2] >= best[2]:
            best, choices = r, trial
    return best


# ---------------------------------------------------------------------------------------------------------------- big-int ZDAU (field values)
def zdau_field(p, x1, y1, x2, y2, z):
    """curve_group.h:120-153 on integers mod p (the values, whatever the representation)."""
    dx = (x1 - x2) % p; Cp = dx * dx % p; W1p = x1 * Cp % p; W2p = x2 * Cp % p
    dy = (y1 - y2) % p; Dp = dy * dy % p; A1p = y1 * (W1p - W2p) % p
    X3 = (Dp - W1p - W2p) % p; u = (X3 - W1p) % p; Cc = u * u % p
    yp = ((dy - u) ** 2 - Dp - Cc) % p
    Y3p = (yp - 2 * A1p) % p; ym = (Y3p - 2 * A1p) % p
    W1 = 4 * X3 * Cc % p; W2 = 4 * W1p * Cc % p
    A1 = Y3p * (W1 - W2) % p
    zz = ((dx + u) ** 2 - Cp - Cc) % p
    z3 = z * zz % p
    D = ym * ym % p; nx1 = (D - W1 - W2) % p; ny1 = (ym * (W1 - nx1) - A1) % p
    Dc = yp * yp % p; nx2 = (Dc - W1 - W2) % p; ny2 = (yp * (W1 - nx2) - A1) % p
    return nx1, ny1, nx2, ny2, z3


if __name__ == "__main__":
    r = prove_invariant(CURVE_ANY)
    print(CURVE_ANY.name, "ladder invariant holds; worst column 2^%d (%.6f of 2^63), worst limb 2^%d" % (r["worst_column_bits"], r["worst_column"] / 2**63, r["worst_limb_bits"]))
    for cv in (CURVE_P256, CURVE_SECP):
        r = prove_invariant(cv)
        print(cv.name, "invariant holds; worst column 2^%d, worst limb 2^%d" % (r["worst_column_bits"], r["worst_limb_bits"]))
        for k, v in r["out"].items():
            print("  ", k, v)
        c = prove_comb_invariant(cv)
        print(cv.name, "comb invariant holds; worst column 2^%d, worst limb 2^%d" % (c["worst_column_bits"], c["worst_limb_bits"]))
        c = prove_window_invariant(cv)
        print(cv.name, "window-loop invariant holds; worst column 2^%d, worst limb 2^%d" % (c["worst_column_bits"], c["worst_limb_bits"]))
    c = prove_complete_invariant(CURVE_SECP)
    print("secp256k1 complete-addition invariant holds; worst column 2^%d, worst limb 2^%d" % (c["worst_column_bits"], c["worst_limb_bits"]))
