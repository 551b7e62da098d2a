#!/usr/bin/env python3
"""Soak run on the GPU alone: the three ways this library computes an affine k*P must agree lane for lane --
the reference ladder (+ simultaneous inversion), the per-element-table windowed path (ALG_WINDOWED, and its constant-time form) and, for
P = G, the three window-table kernels and the constant-time form of the 4-bit one; and the x-only products (on P-256 the ladder without Z) must give the same x.  Different algorithms over the same field layer: a disagreement means a bug
in one of them.  Usage: soak_windowed.py [lanes_per_batch_log2=22] [batches=8] [registered curves, comma-separated]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from ecsimd_amd import Engine, P256, SECP256K1, OUT_AFFINE, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG, ALG_CONSTANT_TIME
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
e = Engine(0); n = 1 << log2n
tot = bad = 0
t0 = time.time()
for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    for b in range(batches):
        seed = 0x5EED0000 + 131 * b + cv
        k = e.fill_random(n, seed, 1); s = e.fill_random(n, seed, 2)
        g4 = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED)
        g7 = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED_SIGNED)
        g16 = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED_BIG)
        gct = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME)
        gl = e.scalar_mult_base(cv, s, flags=OUT_AFFINE)
        d_fixed = int(((g4[0] != gl[0]).any(dim=1) | (g4[1] != gl[1]).any(dim=1) | (g7[0] != gl[0]).any(dim=1) | (g7[1] != gl[1]).any(dim=1)
                       | (g16[0] != gl[0]).any(dim=1) | (g16[1] != gl[1]).any(dim=1) | (gct[0] != gl[0]).any(dim=1) | (gct[1] != gl[1]).any(dim=1)).sum())
        w = e.scalar_mult(cv, k, gl[0], gl[1], flags=OUT_AFFINE | ALG_WINDOWED)
        l = e.scalar_mult(cv, k, gl[0], gl[1], flags=OUT_AFFINE)
        wct = e.scalar_mult(cv, k, gl[0], gl[1], flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME)      # every table entry read, lane masks
        d_var = int(((w[0] != l[0]).any(dim=1) | (w[1] != l[1]).any(dim=1) | (wct[0] != l[0]).any(dim=1) | (wct[1] != l[1]).any(dim=1)).sum())
        del wct
        xo = e.scalar_mult(cv, k, gl[0], gl[1], flags=OUT_AFFINE, x_only=True)[0]      # P-256: the ladder without Z
        xg = e.scalar_mult_base(cv, s, flags=OUT_AFFINE, x_only=True)[0]
        d_x = int(((xo != l[0]).any(dim=1) | (xg != gl[0]).any(dim=1)).sum())
        tot += 3 * n; bad += d_fixed + d_var + d_x
        print(f"{nm} batch {b}: {n} fixed-base + {n} variable-base + {n} x-only lanes, differing: {d_fixed} / {d_var} / {d_x}   [{time.time()-t0:.0f}s]", flush=True)
# (r5) curves registered at run time: the generator's comb (plain and constant-time) against the ladder, u1 G + u2 Q (the comb + the window loop + an affine
# addition) against two ladder passes and the same addition, and the variable-base window loop against the ladder -- argv[3] = comma-separated names, e.g. brainpoolP256r1,sm2,frp256v1
for nm in (sys.argv[3].split(",") if len(sys.argv) > 3 else []):
    from ecsimd_amd.curves import curve_id
    cv = curve_id(nm)
    for b in range(batches):
        seed = 0x5EED0000 + 131 * b + (cv & 0xff) + 77
        k = e.fill_random(n, seed, 1, clear_top_bits=1); s = e.fill_random(n, seed, 2, clear_top_bits=1)
        gl = e.scalar_mult_base(cv, s, flags=OUT_AFFINE)
        g4 = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED)
        gct = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME)
        d_fixed = int(((g4[0] != gl[0]).any(dim=1) | (g4[1] != gl[1]).any(dim=1) | (gct[0] != gl[0]).any(dim=1) | (gct[1] != gl[1]).any(dim=1)).sum())
        del g4, gct
        dx, dy, fin = e.double_scalar_mult(cv, s, k, gl[0], gl[1])                         # s G + k (s G)
        l = e.scalar_mult(cv, k, gl[0], gl[1], flags=OUT_AFFINE)
        ax, ay, afin = e.affine_add(cv, gl, l)
        d_var = int(((dx != ax).any(dim=1) | (dy != ay).any(dim=1) | (fin != afin)).sum())
        w = e.scalar_mult(cv, k, gl[0], gl[1], flags=OUT_AFFINE | ALG_WINDOWED)            # the lane's own window table (k_gvarwin.hip) against the ladder
        d_win = int(((w[0] != l[0]).any(dim=1) | (w[1] != l[1]).any(dim=1)).sum())
        del w
        tot += 3 * n; bad += d_fixed + d_var + d_win
        print(f"{nm} batch {b}: {n} fixed-base (comb, constant-time comb vs ladder) + {n} u1 G + u2 Q + {n} variable-base window-loop lanes, differing: {d_fixed} / {d_var} / {d_win}   [{time.time()-t0:.0f}s]", flush=True)
print(f"TOTAL {tot} scalar multiplications compared across algorithms, {bad} lanes differ")
sys.exit(1 if bad else 0)
