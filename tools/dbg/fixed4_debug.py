"""Which lanes of the 4-bit LDS fixed-base kernel differ from the ladder?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ecsimd_amd import Engine, P256, SECP256K1, OUT_AFFINE, ALG_WINDOWED, ALG_WINDOWED_SIGNED
from helpers import fill_random_np, SEED, ints_to_arr, to_int, CURVE_PARAMS
e = Engine(0)
for cv in (P256, SECP256K1):
    order = CURVE_PARAMS[cv]["n"]
    n = 4096
    k = fill_random_np(n, SEED, 5)
    edge = [1, 2, 3, 15, 16, 17, 255, 256, 2**64 - 1, 2**64, 2**128 + 1, 2**252, 15 * 2**252, order - 2, order + 1, order + 2, 2**256 - 1, 0, order, order - 1]
    k[:len(edge)] = ints_to_arr(edge)
    dk = e.to_device(k)
    wx, wy = e.scalar_mult_base(cv, dk, flags=OUT_AFFINE | ALG_WINDOWED)
    sx, sy = e.scalar_mult_base(cv, dk, flags=OUT_AFFINE | ALG_WINDOWED_SIGNED)
    bad = torch.nonzero(((wx != sx) | (wy != sy)).any(dim=1)).flatten().cpu().numpy()
    print("curve", cv, "differing lanes:", len(bad), bad[:20])
    for i in bad[:12]:
        print("  lane", i, hex(to_int(k[i])), " x equal:", bool((wx[i] == sx[i]).all()), " y equal:", bool((wy[i] == sy[i]).all()))
