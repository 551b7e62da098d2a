#!/usr/bin/env python3
"""Soak parity run: GPU ladder vs the REAL reference (oracle/_ref) on many seeded batches, both curves.
Every differing lane must be explained by the reference's square() defect (exact oracle == GPU and
faithful oracle == reference).  Usage: soak.py [lanes_per_curve_log2=22] [batches=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from ecsimd_amd import Engine, P256, SECP256K1
from oracle import loader
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import usable_cores
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
e = Engine(0); ref = loader.Reference(); ex = loader.Oracle(False); fa = loader.Oracle(True)
cores = usable_cores(); n = 1 << log2n
tot = diff = unexplained = 0
t0 = time.time()
for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    for b in range(batches):
        seed = 0xC0FFEE00 + 977 * b + cv
        k = e.fill_random(n, seed, 1); s = e.fill_random(n, seed, 2)
        bx, by = e.scalar_mult_base(cv, s, flags=6)                      # windowed path makes the base points
        J = e.scalar_mult(cv, k, bx, by)
        kn, xn, yn = (e.to_numpy(t) for t in (k, bx, by)); g = [e.to_numpy(t) for t in J]
        r = ref.scalar_mult(cv, kn, xn, yn, threads=cores)
        bad = np.nonzero((g[0] != r[0]).any(axis=1) | (g[1] != r[1]).any(axis=1) | (g[2] != r[2]).any(axis=1))[0]
        ok = True
        if len(bad):
            e_ = ex.scalar_mult(cv, kn[bad], xn[bad], yn[bad], threads=min(cores, len(bad)))
            f_ = fa.scalar_mult(cv, kn[bad], xn[bad], yn[bad], threads=min(cores, len(bad)))
            ok = all(np.array_equal(u, v[bad]) for u, v in zip(e_, g)) and all(np.array_equal(u, v[bad]) for u, v in zip(f_, r))
        tot += n; diff += len(bad); unexplained += 0 if ok else len(bad)
        print(f"{nm} batch {b}: {n} lanes, {len(bad)} differ from the reference, explained by its square() defect: {ok}   [{time.time()-t0:.0f}s]", flush=True)
print(f"TOTAL {tot} scalar multiplications, {diff} lanes differ ({diff/tot:.2e}), unexplained: {unexplained}")
sys.exit(1 if unexplained else 0)
