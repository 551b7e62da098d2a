"""ctypes front-ends for the two parity checkers.  TEST INFRASTRUCTURE ONLY.

* ``Oracle``      -> oracle/libecsimd_oracle.so  (C restatement, ecsimd_oracle.c)
* ``OpenSSLCheck`` -> oracle/libecsimd_ossl.so   (libcrypto; independent level-A cross-check and the
                     reference's competitor benchmark, ossl_check.c)
* ``Reference``   -> oracle/_ref/libecsimd_ref.so (the real aguinet/ecsimd headers behind ref_driver.cpp;
                     present only where oracle/Makefile could build it, i.e. the build container, and
                     shipped to the GPU box as a prebuilt artefact)

Both expose the same method names over numpy ``uint64`` arrays of shape (n, 4) (little-endian limb
order; 512-bit values are (n, 8)), so a test can run the same call against either and compare.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
P256, SECP256K1 = 0, 1
CURVES = {"p256": P256, "secp256k1": SECP256K1}

_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def _p(a):
    return None if a is None else a.ctypes.data_as(_u64p)


def _p8(a):
    return None if a is None else a.ctypes.data_as(_u8p)


def _arr(a, words=4):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim == 2 and a.shape[1] == words, a.shape
    return a


def build(force: bool = False) -> None:
    """Compile the C restatement (always) and the reference driver (when /root/reference exists)."""
    args = ["make", "-C", HERE, "-s"] + (["-B"] if force else []) + ["all"]
    subprocess.run(args, check=True)


class _Lib:
    prefix = ""
    path = ""

    def __init__(self):
        if not os.path.exists(self.path):
            raise FileNotFoundError(self.path)
        self.lib = C.CDLL(self.path)
        self.now = getattr(self.lib, self.prefix + "now")
        self.now.restype = C.c_double

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    # ---- constants
    def constants(self, curve):
        out = np.zeros((12, 4), dtype=np.uint64)
        mp = C.c_uint32(0)
        rc = self._f("get_constants")(C.c_int(curve), _p(out), C.byref(mp))
        assert rc == 0
        names = ["p", "a", "b", "gx", "gy", "r_p", "rsq_p", "pm1_r_p", "am", "bm", "p_m2", "p_sqrt"]
        d = {k: out[i].copy() for i, k in enumerate(names)}
        d["mprime"] = mp.value
        return d

    # ---- curve-independent bignum ops
    def _bin_flag(self, name, a, b):
        a, b = _arr(a), _arr(b)
        out = np.empty_like(a)
        flag = np.zeros(len(a), dtype=np.uint8)
        assert self._f(name)(_p(a), _p(b), _p(out), _p8(flag), C.c_size_t(len(a))) == 0
        return out, flag

    def add(self, a, b):
        return self._bin_flag("add", a, b)

    def sub(self, a, b):
        return self._bin_flag("sub", a, b)

    def sub_if_above(self, a, p):
        a, p = _arr(a), _arr(p)
        out = np.empty_like(a)
        assert self._f("sub_if_above")(_p(a), _p(p), _p(out), C.c_size_t(len(a))) == 0
        return out

    def shift_left_one(self, a):
        a = _arr(a)
        out = np.empty_like(a)
        flag = np.zeros(len(a), dtype=np.uint8)
        assert self._f("shift_left_one")(_p(a), _p(out), _p8(flag), C.c_size_t(len(a))) == 0
        return out, flag

    def mul(self, a, b):
        a, b = _arr(a), _arr(b)
        out = np.empty((len(a), 8), dtype=np.uint64)
        assert self._f("mul")(_p(a), _p(b), _p(out), C.c_size_t(len(a))) == 0
        return out

    def square(self, a):
        a = _arr(a)
        out = np.empty((len(a), 8), dtype=np.uint64)
        assert self._f("square")(_p(a), _p(out), C.c_size_t(len(a))) == 0
        return out

    # ---- field ops
    def _c2(self, name, curve, a, b):
        a, b = _arr(a), _arr(b)
        out = np.empty_like(a)
        assert self._f(name)(C.c_int(curve), _p(a), _p(b), _p(out), C.c_size_t(len(a))) == 0
        return out

    def _c1(self, name, curve, a, words=4):
        a = _arr(a, words)
        out = np.empty((len(a), 4), dtype=np.uint64)
        assert self._f(name)(C.c_int(curve), _p(a), _p(out), C.c_size_t(len(a))) == 0
        return out

    def mod_add(self, curve, a, b):
        return self._c2("mod_add", curve, a, b)

    def mod_sub(self, curve, a, b):
        return self._c2("mod_sub", curve, a, b)

    def mod_shift_left(self, curve, a, count):
        a = _arr(a)
        out = np.empty_like(a)
        assert self._f("mod_shift_left")(C.c_int(curve), _p(a), C.c_int(count), _p(out), C.c_size_t(len(a))) == 0
        return out

    def mgry_reduce(self, curve, a8):
        return self._c1("mgry_reduce", curve, a8, 8)

    def mgry_mul(self, curve, a, b):
        return self._c2("mgry_mul", curve, a, b)

    def mgry_sqr(self, curve, a):
        return self._c1("mgry_sqr", curve, a)

    def mgry_from_classical(self, curve, a):
        return self._c1("mgry_from_classical", curve, a)

    def mgry_to_classical(self, curve, a):
        return self._c1("mgry_to_classical", curve, a)

    def mgry_pow(self, curve, a, exponent):
        a = _arr(a)
        e = _arr(np.asarray(exponent, dtype=np.uint64).reshape(1, 4))
        out = np.empty_like(a)
        assert self._f("mgry_pow")(C.c_int(curve), _p(a), _p(e), _p(out), C.c_size_t(len(a))) == 0
        return out

    def gfp_inverse(self, curve, a):
        return self._c1("gfp_inverse", curve, a)

    def gfp_opposite(self, curve, a):
        return self._c1("gfp_opposite", curve, a)

    def gfp_sqrt(self, curve, a):
        a = _arr(a)
        out = np.zeros_like(a)
        ok = np.zeros(len(a), dtype=np.uint8)
        assert self._f("gfp_sqrt")(C.c_int(curve), _p(a), _p(out), _p8(ok), C.c_size_t(len(a))) == 0
        return out, ok

    # ---- points.  In-out arguments are copied first; the updated copies are returned.
    def dblu(self, curve, p):
        px, py, pz = (_arr(v).copy() for v in p)
        r = [np.empty_like(px) for _ in range(3)]
        assert self._f("dblu")(C.c_int(curve), _p(px), _p(py), _p(pz), _p(r[0]), _p(r[1]), _p(r[2]), C.c_size_t(len(px))) == 0
        return tuple(r), (px, py, pz)

    def trplu(self, curve, p):
        px, py, pz = (_arr(v).copy() for v in p)
        r = [np.empty_like(px) for _ in range(3)]
        assert self._f("trplu")(C.c_int(curve), _p(px), _p(py), _p(pz), _p(r[0]), _p(r[1]), _p(r[2]), C.c_size_t(len(px))) == 0
        return tuple(r), (px, py, pz)

    def zaddu(self, curve, p, o):
        px, py, pz = (_arr(v).copy() for v in p)
        ox, oy, oz = (_arr(v) for v in o)
        r = [np.empty_like(px) for _ in range(3)]
        assert self._f("zaddu")(C.c_int(curve), _p(px), _p(py), _p(pz), _p(ox), _p(oy), _p(oz), _p(r[0]), _p(r[1]), _p(r[2]), C.c_size_t(len(px))) == 0
        return tuple(r), (px, py, pz)

    def zdau(self, curve, p, q):
        px, py, pz = (_arr(v) for v in p)
        qx, qy, qz = (_arr(v).copy() for v in q)
        r = [np.empty_like(px) for _ in range(3)]
        assert self._f("zdau")(C.c_int(curve), _p(px), _p(py), _p(pz), _p(qx), _p(qy), _p(qz), _p(r[0]), _p(r[1]), _p(r[2]), C.c_size_t(len(px))) == 0
        return tuple(r), (qx, qy, qz)

    def add_z2_1(self, curve, a, bxy):
        ax, ay, az = (_arr(v) for v in a)
        bx, by = (_arr(v) for v in bxy)
        r = [np.empty_like(ax) for _ in range(3)]
        assert self._f("add_z2_1")(C.c_int(curve), _p(ax), _p(ay), _p(az), _p(bx), _p(by), _p(r[0]), _p(r[1]), _p(r[2]), C.c_size_t(len(ax))) == 0
        return tuple(r)

    def from_affine(self, curve, x, y):
        x, y = _arr(x), _arr(y)
        r = [np.empty_like(x) for _ in range(3)]
        assert self._f("from_affine")(C.c_int(curve), _p(x), _p(y), _p(r[0]), _p(r[1]), _p(r[2]), C.c_size_t(len(x))) == 0
        return tuple(r)

    def to_affine(self, curve, j):
        jx, jy, jz = (_arr(v) for v in j)
        x, y = np.empty_like(jx), np.empty_like(jx)
        assert self._f("to_affine")(C.c_int(curve), _p(jx), _p(jy), _p(jz), _p(x), _p(y), C.c_size_t(len(jx))) == 0
        return x, y

    def compute_y(self, curve, x):
        x = _arr(x)
        y = np.zeros_like(x)
        ok = np.zeros(len(x), dtype=np.uint8)
        assert self._f("compute_y")(C.c_int(curve), _p(x), _p(y), _p8(ok), C.c_size_t(len(x))) == 0
        return y, ok

    def scalar_mult(self, curve, k, x, y, threads=1, mgry_in=False):
        k, x, y = _arr(k), _arr(x), _arr(y)
        r = [np.empty_like(k) for _ in range(3)]
        name = "scalar_mult_mgry" if mgry_in else "scalar_mult"
        assert self._f(name)(C.c_int(curve), _p(k), _p(x), _p(y), _p(r[0]), _p(r[1]), _p(r[2]), C.c_size_t(len(k)), C.c_int(threads)) == 0
        return tuple(r)


    def bench_ops(self, curve, op, a, b, iters):
        """BASELINE configs[0] (benchs/ops.cpp) as a timed loop inside the library: op 0 mgry_sqr_256, 1 mgry_reduce_512
        (a = 512-bit values), 2 mul_256.  Returns (seconds for `iters` passes over the len(a) elements, last results)."""
        a = _arr(a, 8 if op == 1 else 4)
        b = _arr(b) if b is not None else a
        out = np.zeros((len(a), 8 if op == 2 else 4), dtype=np.uint64)
        f = self._f("bench_ops"); f.restype = C.c_double
        dt = f(C.c_int(curve), C.c_int(op), _p(a), _p(b), _p(out), C.c_size_t(len(a)), C.c_size_t(iters))
        assert dt >= 0, dt
        return float(dt), out


class Oracle(_Lib):
    """The C restatement.  ``faithful=False`` (default): exact squaring, the arithmetic the reference
    specifies -- what the HIP path is checked against.  ``faithful=True``: bug-for-bug restatement of
    the reference's square() including its dropped carry (ecsimd_oracle.c, bn_square)."""
    prefix = "oracle_"
    path = os.path.join(HERE, "libecsimd_oracle.so")

    def __init__(self, faithful: bool = False):
        super().__init__()
        self.faithful = bool(faithful)
        self.lib.oracle_dropped_carries.restype = C.c_ulonglong

    def _f(self, name):
        self.lib.oracle_set_square_mode(C.c_int(1 if self.faithful else 0))   # one shared library: select per call
        return super()._f(name)

    def register_modulus(self, p: int) -> int:
        """Field id (>= 2) of the odd modulus p, for the mod_* / mgry_* / gfp_* methods (not the point methods)."""
        limbs = np.array([(p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        fid = int(self.lib.oracle_register_modulus(_p(limbs)))
        assert fid >= 2, (fid, hex(p))
        return fid

    def register_curve(self, p: int, a: int, b: int, gx: int, gy: int) -> int:
        """Curve id of y^2 = x^3 + a x + b over GF(p) with generator (gx, gy), for every method (the point methods included)."""
        lim = lambda v: np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        cid = int(self.lib.oracle_register_curve(_p(lim(p)), _p(lim(a)), _p(lim(b)), _p(lim(gx)), _p(lim(gy))))
        assert cid >= 2, (cid, hex(p))
        return cid

    def dropped_carries(self) -> int:
        return int(self.lib.oracle_dropped_carries())

    def reset_dropped_carries(self) -> None:
        self.lib.oracle_reset_dropped_carries()


# The curve-less moduli oracle/ref_driver.cpp instantiates the reference's field layer for (field ids 2..6 there): the two group
# orders, a prime below 2^255, the largest odd 256-bit value (composite) and a 192-bit prime.  Only the ones = 3 mod 4 have a
# GFp<WBN, P> in the reference (gfp.h:84): sqrt / opposite exist for those alone.
REF_MODULI = {
    "n_p256": 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551,
    "n_secp256k1": 0xfffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141,
    "p25519": 2 ** 255 - 19,
    "all_ones": 2 ** 256 - 1,
    "p192": 2 ** 192 - 2 ** 64 - 1,
}


# The curves oracle/ref_driver.cpp instantiates curve_group<Curve> for beside P-256 and secp256k1 (its curve ids 10, 11, 12): public parameters
# (RFC 5639 3.4; GB/T 32918.5 / RFC 8998; ANSSI FRP256v1), every p = 3 mod 4 as the reference's GFp needs.  n = the group order (the reference has none).
REF_CURVES = {
    "brainpoolP256r1": dict(ref_id=10,
        p=0xa9fb57dba1eea9bc3e660a909d838d726e3bf623d52620282013481d1f6e5377, a=0x7d5a0975fc2c3057eef67530417affe7fb8055c126dc5c6ce94a4b44f330b5d9,
        b=0x26dc5c6ce94a4b44f330b5d9bbd77cbf958416295cf7e1ce6bccdc18ff8c07b6, gx=0x8bd2aeb9cb7e57cb2c4b482ffc81b7afb9de27e1e3bd23c23a4453bd9ace3262,
        gy=0x547ef835c3dac4fd97f8461a14611dc9c27745132ded8e545c1d54c72f046997, n=0xa9fb57dba1eea9bc3e660a909d838d718c397aa3b561a6f7901e0e82974856a7),
    "sm2": dict(ref_id=11,
        p=0xfffffffeffffffffffffffffffffffffffffffff00000000ffffffffffffffff, a=0xfffffffeffffffffffffffffffffffffffffffff00000000fffffffffffffffc,
        b=0x28e9fa9e9d9f5e344d5a9e4bcf6509a7f39789f515ab8f92ddbcbd414d940e93, gx=0x32c4ae2c1f1981195f9904466a39c9948fe30bbff2660be1715a4589334c74c7,
        gy=0xbc3736a2f4f6779c59bdcee36b692153d0a9877cc62a474002df32e52139f0a0, n=0xfffffffeffffffffffffffffffffffff7203df6b21c6052b53bbf40939d54123),
    "frp256v1": dict(ref_id=12,
        p=0xf1fd178c0b3ad58f10126de8ce42435b3961adbcabc8ca6de8fcf353d86e9c03, a=0xf1fd178c0b3ad58f10126de8ce42435b3961adbcabc8ca6de8fcf353d86e9c00,
        b=0xee353fca5428a9300d4aba754a44c00fdfec0c9ae4b1a1803075ed967b7bb73f, gx=0xb6b3d4c356c139eb31183d4749d423958c27d2dcaf98b70164c97a2dd98f5cff,
        gy=0x6142e0f7c8b204911f9271f0f3ecef8c2701c307e8e4c9e183115a1554062cfb, n=0xf1fd178c0b3ad58f10126de8ce42435b53dc67e140d2bf941ffdd459c6d655e1),
}


class Reference(_Lib):
    prefix = "ref_"
    path = os.path.join(HERE, "_ref", "libecsimd_ref.so")

    def register_curve(self, p, a, b, gx, gy) -> int:
        """The reference is generic at COMPILE time: only the curves of REF_CURVES (and the two built-in ones) have an instance."""
        for c in REF_CURVES.values():
            if (c["p"], c["a"], c["b"], c["gx"], c["gy"]) == (p, a, b, gx, gy):
                return c["ref_id"]
        raise KeyError(hex(p))

    def register_modulus(self, p: int) -> int:
        """The reference is generic at COMPILE time: only the moduli of REF_MODULI have an instance (ids 2..6)."""
        for fid in range(2, 2 + len(REF_MODULI)):
            if sum(int(v) << (64 * i) for i, v in enumerate(self.constants(fid)["p"])) == p:
                return fid
        raise KeyError(hex(p))

    def scalar_mult_1s(self, curve, k1, x, y):
        k1 = _arr(np.asarray(k1, dtype=np.uint64).reshape(1, 4))
        x, y = _arr(x), _arr(y)
        r = [np.empty_like(x) for _ in range(3)]
        assert self._f("scalar_mult_1s")(C.c_int(curve), _p(k1), _p(x), _p(y), _p(r[0]), _p(r[1]), _p(r[2]), C.c_size_t(len(x))) == 0
        return tuple(r)


class OpenSSLCheck:
    """oracle/libecsimd_ossl.so (ossl_check.c): P-256 / secp256k1 point multiplication through OpenSSL's
    libcrypto -- an implementation independent of both the reference and the restatement; affine classical
    coordinates in and out, ``inf`` flags the point at infinity."""
    path = os.path.join(HERE, "libecsimd_ossl.so")

    def __init__(self):
        if not os.path.exists(self.path):
            raise FileNotFoundError(self.path)
        self.lib = C.CDLL(self.path)
        self.lib.ossl_time_scalar_mult.restype = C.c_double
        self.lib.ossl_version.restype = C.c_char_p

    def version(self) -> str:
        return self.lib.ossl_version().decode()

    def _out(self, n):
        return np.empty((n, 4), dtype=np.uint64), np.empty((n, 4), dtype=np.uint64), np.zeros(n, dtype=np.uint8)

    def scalar_mult(self, curve, k, x, y, threads=1):
        k, x, y = _arr(k), _arr(x), _arr(y)
        ox, oy, inf = self._out(len(k))
        rc = self.lib.ossl_scalar_mult(C.c_int(curve), _p(k), _p(x), _p(y), _p(ox), _p(oy), _p8(inf), C.c_size_t(len(k)), C.c_int(threads))
        assert rc == 0, rc
        return ox, oy, inf

    def scalar_mult_base(self, curve, k, threads=1):
        k = _arr(k)
        ox, oy, inf = self._out(len(k))
        rc = self.lib.ossl_scalar_mult_base(C.c_int(curve), _p(k), _p(ox), _p(oy), _p8(inf), C.c_size_t(len(k)), C.c_int(threads))
        assert rc == 0, rc
        return ox, oy, inf

    def double_scalar_mult(self, curve, u1, u2, qx, qy, threads=1):
        u1, u2, qx, qy = _arr(u1), _arr(u2), _arr(qx), _arr(qy)
        ox, oy, inf = self._out(len(u1))
        rc = self.lib.ossl_double_scalar_mult(C.c_int(curve), _p(u1), _p(u2), _p(qx), _p(qy), _p(ox), _p(oy), _p8(inf),
                                              C.c_size_t(len(u1)), C.c_int(threads))
        assert rc == 0, rc
        return ox, oy, inf

    def ecdsa_sign(self, curve, d, e, threads=1):
        """Key pairs from the private keys d and signatures of the digests e (as integers): returns r, s, qx, qy."""
        d, e = _arr(d), _arr(e)
        r, s, qx, qy = (np.empty_like(d) for _ in range(4))
        rc = self.lib.ossl_ecdsa_sign(C.c_int(curve), _p(d), _p(e), _p(r), _p(s), _p(qx), _p(qy), C.c_size_t(len(d)), C.c_int(threads))
        assert rc == 0, rc
        return r, s, qx, qy

    def ecdsa_verify(self, curve, e, r, s, qx, qy, threads=1):
        """ECDSA_do_verify per element: 1 = accepted; bad signatures, invalid public keys, out-of-range r / s give 0."""
        e, r, s, qx, qy = _arr(e), _arr(r), _arr(s), _arr(qx), _arr(qy)
        ok = np.zeros(len(e), dtype=np.uint8)
        rc = self.lib.ossl_ecdsa_verify(C.c_int(curve), _p(e), _p(r), _p(s), _p(qx), _p(qy), _p8(ok), C.c_size_t(len(e)), C.c_int(threads))
        assert rc == 0, rc
        return ok

    def time_scalar_mult(self, curve, k, x, y, threads=1) -> float:
        """Seconds for len(k) variable-base multiplications on `threads` threads (benchs/p256_ref.cpp:55-91)."""
        k, x, y = _arr(k), _arr(x), _arr(y)
        t = self.lib.ossl_time_scalar_mult(C.c_int(curve), _p(k), _p(x), _p(y), C.c_size_t(len(k)), C.c_int(threads))
        assert t > 0, t
        return float(t)


def openssl_available() -> bool:
    return os.path.exists(OpenSSLCheck.path)


def reference_available() -> bool:
    return os.path.exists(Reference.path)


# ---------------------------------------------------------------- helpers shared by tests
def to_int(limbs) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(limbs))


def from_int(v: int, words: int = 4) -> np.ndarray:
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(words)], dtype=np.uint64)


def ints_to_arr(vals, words: int = 4) -> np.ndarray:
    return np.stack([from_int(int(v), words) for v in vals]) if len(vals) else np.zeros((0, words), dtype=np.uint64)


def arr_to_ints(a):
    return [to_int(row) for row in np.asarray(a)]


def from_hex(h: str, words: int = 4) -> np.ndarray:
    """serialization.h:12-24 bn_from_bytes_BE on a big-endian hex literal."""
    return from_int(int(h, 16), words)
