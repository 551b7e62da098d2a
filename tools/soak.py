#!/usr/bin/env python3
"""Soak parity run: GPU ladder vs the REAL reference (oracle/_ref) on many seeded batches, both curves.

Default ladder (exact squaring): every lane that differs from the reference must be explained by the reference's
square() defect (exact oracle == GPU and faithful oracle == reference) AND confirmed by libcrypto, which shares no code
or algorithm with either: OpenSSL's k*P must equal the GPU's affine point on that lane and differ from the reference's.
With ECSIMD_HIP_REF_SQUARE_COMPAT (the second pass over the same inputs) not one lane may differ.
Curves registered at run time (round 5) run against the reference instantiated for them (oracle/ref_driver.cpp ids 10 / 11 / 12); libcrypto's harness
knows the two built-in curves only, so there a differing lane is settled by textbook affine arithmetic on Python integers instead.
Usage: soak.py [lanes_per_batch_log2=22] [batches=2] [curves=p256,secp256k1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from ecsimd_amd import Engine, P256, SECP256K1, OUT_AFFINE, ALG_WINDOWED_BIG, REF_SQUARE_COMPAT
from oracle import loader
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import usable_cores, openssl_checker
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
names = sys.argv[3].split(",") if len(sys.argv) > 3 else ["p256", "secp256k1"]
e = Engine(0); ref = loader.Reference(); ex = loader.Oracle(False); fa = loader.Oracle(True); ossl = openssl_checker()
cores = usable_cores(); n = 1 << log2n
tot = diff = unexplained = confirmed = compat_diff = 0
t0 = time.time()


def textbook(c, k, x, y):
    """k (x, y) on y^2 = x^3 + a x + b over GF(p) by affine double-and-add on Python integers (None = infinity)."""
    p, a = c["p"], c["a"]
    def add(P, Q):
        if P is None: return Q
        if Q is None: return P
        if P[0] == Q[0]:
            if (P[1] + Q[1]) % p == 0: return None
            lam = (3 * P[0] * P[0] + a) * pow(2 * P[1], -1, p) % p
        else:
            lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
        x3 = (lam * lam - P[0] - Q[0]) % p
        return x3, (lam * (P[0] - x3) - P[1]) % p
    R = None
    for bit in bin(k)[2:] if k else "":
        R = add(R, R)
        if bit == "1": R = add(R, (x, y))
    return R


for nm in names:
    reg = nm not in ("p256", "secp256k1")
    if reg:
        from ecsimd_amd.curves import curve_id
        cv = curve_id(nm); rid = loader.REF_CURVES[nm]["ref_id"]
        oid_e, oid_f = (o.register_curve(*(loader.REF_CURVES[nm][key] for key in ("p", "a", "b", "gx", "gy"))) for o in (ex, fa))
    else:
        cv = rid = oid_e = oid_f = P256 if nm == "p256" else SECP256K1
    for b in range(batches):
        seed = 0xC0FFEE00 + 977 * b + (cv & 0xffff) + (31 if reg else 0)
        k = e.fill_random(n, seed, 1); s = e.fill_random(n, seed, 2)
        bx, by = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | (0 if reg else ALG_WINDOWED_BIG))      # a windowed path makes the base points (registered curves: the ladder)
        J = e.scalar_mult(cv, k, bx, by)
        Jc = e.scalar_mult(cv, k, bx, by, flags=REF_SQUARE_COMPAT)
        kn, xn, yn = (e.to_numpy(t) for t in (k, bx, by)); g = [e.to_numpy(t) for t in J]; gc = [e.to_numpy(t) for t in Jc]
        r = ref.scalar_mult(rid, kn, xn, yn, threads=cores)
        bad = np.nonzero((g[0] != r[0]).any(axis=1) | (g[1] != r[1]).any(axis=1) | (g[2] != r[2]).any(axis=1))[0]
        cbad = int(np.count_nonzero((gc[0] != r[0]).any(axis=1) | (gc[1] != r[1]).any(axis=1) | (gc[2] != r[2]).any(axis=1)))
        ok, conf = True, 0
        if len(bad):
            e_ = ex.scalar_mult(oid_e, kn[bad], xn[bad], yn[bad], threads=min(cores, len(bad)))
            f_ = fa.scalar_mult(oid_f, kn[bad], xn[bad], yn[bad], threads=min(cores, len(bad)))
            ok = all(np.array_equal(u, v[bad]) for u, v in zip(e_, g)) and all(np.array_equal(u, v[bad]) for u, v in zip(f_, r))
            if reg:
                c = loader.REF_CURVES[nm]; ti = lambda v: sum(int(w) << (64 * j) for j, w in enumerate(v))
                ax, ay = (e.to_numpy(t) for t in e.to_affine(cv, [e.select_rows(t, bad) for t in J]))
                rx, ry = ref.to_affine(rid, [v[bad] for v in r])
                for j, lane in enumerate(bad):
                    want = textbook(c, ti(kn[lane]), ti(xn[lane]), ti(yn[lane]))
                    conf += int(want == (ti(ax[j]), ti(ay[j])) and want != (ti(rx[j]), ti(ry[j])))
            elif ossl is not None:
                ax, ay = (e.to_numpy(t) for t in e.to_affine(cv, [e.select_rows(t, bad) for t in J]))
                vx, vy, inf = ossl.scalar_mult(cv, kn[bad], xn[bad], yn[bad], threads=1)
                rx, ry = ref.to_affine(cv, [v[bad] for v in r])
                gpu_right = ~((ax != vx).any(axis=1) | (ay != vy).any(axis=1) | (inf != 0))
                ref_wrong = (rx != vx).any(axis=1) | (ry != vy).any(axis=1)
                conf = int(np.count_nonzero(gpu_right & ref_wrong))
        tot += n; diff += len(bad); unexplained += 0 if ok else len(bad); confirmed += conf; compat_diff += cbad
        who = "textbook affine arithmetic on Python integers" if reg else "libcrypto"
        print(f"{nm} batch {b}: {n} lanes, {len(bad)} differ from the reference (explained by its square() defect: {ok}; {who} sides with the GPU on {conf}); "
              f"with REF_SQUARE_COMPAT {cbad} differ   [{time.time()-t0:.0f}s]", flush=True)
print(f"TOTAL {tot} scalar multiplications per mode: exact ladder {diff} lanes differ from the reference ({diff/tot:.2e}), unexplained {unexplained}, "
      f"confirmed by OpenSSL (registered curves: by textbook affine arithmetic) {confirmed if ossl is not None else 'n/a'}; REF_SQUARE_COMPAT ladder {compat_diff} lanes differ")
sys.exit(1 if (unexplained or compat_diff or (ossl is not None and confirmed != diff)) else 0)
