import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ecsimd_amd import Engine
from oracle.loader import Oracle, to_int, from_int, ints_to_arr
e = Engine(0); o = Oracle()
rng = np.random.default_rng(1)
pat = np.array([0, 0xffffffff, 0x80000000, 0x7fffffff, 1, 0xfffffffe], dtype=np.uint64)
n = 200000
w = pat[rng.integers(0, len(pat), size=(n, 8))]
rnd = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint64)
use_rnd = rng.integers(0, 4, size=(n, 8)) == 0
w = np.where(use_rnd, rnd, w)
a = (w[:, 0::2] | (w[:, 1::2] << np.uint64(32))).astype(np.uint64)
g2 = e.to_numpy(e.square(e.to_device(a))); x2 = o.square(a)
bad2 = np.nonzero((g2 != x2).any(axis=1))[0]; print("square bad rows", len(bad2), "of", n)
for i in bad2[:12]:
    d = to_int(g2[i]) - to_int(x2[i])
    print(" a=%064x diff=%s%x" % (to_int(a[i]), "-" if d < 0 else "+", abs(d)))
b = np.roll(a, 1, axis=0)
g3 = e.to_numpy(e.mul(e.to_device(a), e.to_device(b))); x3 = o.mul(a, b)
print("mul bad rows", int((g3 != x3).any(axis=1).sum()))
