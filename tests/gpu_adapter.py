"""numpy-in / numpy-out adapter over ecsimd_amd.Engine with the method names of oracle.loader, so
one body of checks (tests/test_oracle.py::run_against_golden) runs against the oracle AND the HIP path."""
import numpy as np


class EngineNP:
    def __init__(self, eng):
        self.e = eng

    def _up(self, a, words=4):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        assert a.ndim == 2 and a.shape[1] == words
        return self.e.to_device(a)

    def _dn(self, t):
        return self.e.to_numpy(t)

    def add(self, a, b):
        s, f = self.e.add(self._up(a), self._up(b)); return self._dn(s), self._dn(f)

    def sub(self, a, b):
        s, f = self.e.sub(self._up(a), self._up(b)); return self._dn(s), self._dn(f)

    def sub_if_above(self, a, p): return self._dn(self.e.sub_if_above(self._up(a), self._up(p)))

    def shift_left_one(self, a):
        s, f = self.e.shift_left_one(self._up(a)); return self._dn(s), self._dn(f)

    def mul(self, a, b): return self._dn(self.e.mul(self._up(a), self._up(b)))
    def square(self, a): return self._dn(self.e.square(self._up(a)))
    def mod_add(self, cv, a, b): return self._dn(self.e.mod_add(cv, self._up(a), self._up(b)))
    def mod_sub(self, cv, a, b): return self._dn(self.e.mod_sub(cv, self._up(a), self._up(b)))
    def mod_mul(self, cv, a, b): return self._dn(self.e.mod_mul(cv, self._up(a), self._up(b)))
    def mod_shift_left(self, cv, a, c): return self._dn(self.e.mod_shift_left(cv, self._up(a), c))
    def mgry_reduce(self, cv, a8): return self._dn(self.e.mgry_reduce(cv, self._up(a8, 8)))
    def mgry_mul(self, cv, a, b): return self._dn(self.e.mgry_mul(cv, self._up(a), self._up(b)))
    def mgry_sqr(self, cv, a): return self._dn(self.e.mgry_sqr(cv, self._up(a)))
    def mgry_from_classical(self, cv, a): return self._dn(self.e.mgry_from_classical(cv, self._up(a)))
    def mgry_to_classical(self, cv, a): return self._dn(self.e.mgry_to_classical(cv, self._up(a)))
    def mgry_pow(self, cv, a, e): return self._dn(self.e.mgry_pow(cv, self._up(a), e))
    def gfp_inverse(self, cv, a): return self._dn(self.e.gfp_inverse(cv, self._up(a)))
    def gfp_opposite(self, cv, a): return self._dn(self.e.gfp_opposite(cv, self._up(a)))

    def gfp_sqrt(self, cv, a):
        s, ok = self.e.gfp_sqrt(cv, self._up(a)); return self._dn(s), self._dn(ok)

    def _pt(self, p): return tuple(self._up(v) for v in p)
    def _ptd(self, p): return tuple(self._dn(v) for v in p)

    def from_affine(self, cv, x, y): return self._ptd(self.e.from_affine(cv, self._up(x), self._up(y)))
    def to_affine(self, cv, j): return self._ptd(self.e.to_affine(cv, self._pt(j)))

    def compute_y(self, cv, x):
        y, ok = self.e.compute_y(cv, self._up(x)); return self._dn(y), self._dn(ok)

    def dblu(self, cv, p):
        dp = self._pt(p); r = self.e.dblu(cv, dp); return self._ptd(r), self._ptd(dp)

    def trplu(self, cv, p):
        dp = self._pt(p); r = self.e.trplu(cv, dp); return self._ptd(r), self._ptd(dp)

    def zaddu(self, cv, p, o):
        dp = self._pt(p); r = self.e.zaddu(cv, dp, self._pt(o)); return self._ptd(r), self._ptd(dp)

    def zdau(self, cv, p, q):
        dq = self._pt(q); r = self.e.zdau(cv, self._pt(p), dq); return self._ptd(r), self._ptd(dq)

    def zdau_repeat(self, cv, p, qxy, iters, swap_bits=0, radix=29):
        return self._ptd(self.e.zdau_repeat(cv, self._pt(p), self._pt(qxy), iters, swap_bits, radix))

    def add_z2_1(self, cv, a, bxy): return self._ptd(self.e.add_z2_1(cv, self._pt(a), self._pt(bxy)))

    def scalar_mult(self, cv, k, x, y, threads=1, mgry_in=False, affine=False, ref_compat=False):
        flags = (1 if mgry_in else 0) | (2 if affine else 0) | (64 if ref_compat else 0)
        return self._ptd(self.e.scalar_mult(cv, self._up(k), self._up(x), self._up(y), flags=flags))

    def scalar_mult_1s(self, cv, k1, x, y): return self._ptd(self.e.scalar_mult_1s(cv, k1, self._up(x), self._up(y)))
    def scalar_mult_base(self, cv, k, affine=False): return self._ptd(self.e.scalar_mult_base(cv, self._up(k), flags=2 if affine else 0))
