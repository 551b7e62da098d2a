"""The one scalar (pair) per comb kernel where its last mixed addition meets R = T: predicted from the recoding, checked against the ladder."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ecsimd_amd import Engine, P256, SECP256K1, OUT_AFFINE, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG
from helpers import ints_to_arr, to_int, CURVE_PARAMS, ec_mul
e = Engine(0)
for cv, name in ((P256, "p256"), (SECP256K1, "secp256k1")):
    c = CURVE_PARAMS[cv]; n = c["n"]; G = (c["gx"], c["gy"])
    ks = []
    for W in (4, 20):
        m = n % (1 << W); ks += [n - 2 * m, 2 * m]
    ks += [n - 2 * (n % (1 << 252)), 2 * (n % (1 << 252)) % n]
    ks += [5, 7]                                                   # controls
    k = e.to_device(ints_to_arr(ks * 64)[:len(ks) * 64])            # a full wave
    lx, ly = e.scalar_mult_base(cv, k, flags=OUT_AFFINE)
    for alg, an in ((ALG_WINDOWED, "4-bit LDS"), (ALG_WINDOWED_SIGNED, "signed 7-bit LDS"), (ALG_WINDOWED_BIG, "20-bit table")):
        wx, wy = e.scalar_mult_base(cv, k, flags=OUT_AFFINE | alg)
        bad = [hex(ks[i]) for i in range(len(ks)) if not (torch.equal(wx[i], lx[i]) and torch.equal(wy[i], ly[i]))]
        exp_ok = all((to_int(e.to_numpy(lx)[i]), to_int(e.to_numpy(ly)[i])) == ec_mul(cv, ks[i], G) for i in range(len(ks)))
        print(name, an, "lanes differing from the ladder:", bad, "| ladder == big-int model:", exp_ok)
