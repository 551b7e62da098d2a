#!/usr/bin/env python3
"""Round 5: the variable-base window loop of a curve registered at run time (k_gvarwin.hip) beside the ladder it is an alternative to.
tools/gvarwin_perf.py [log2 lanes]: scalar_mult(OUT_AFFINE) and scalar_mult(OUT_AFFINE | ALG_WINDOWED) on the three named curves, P-256's parameters through
the generic kernels beside the built-in window loop, and ECDSA verification; HIP-event time of 5 launches after a warm-up, outputs compared lane for lane."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ecsimd_amd import Engine, P256, SECP256K1, ALG_WINDOWED, ALG_CONSTANT_TIME, OUT_AFFINE     # noqa: E402
from ecsimd_amd.engine import register_curve                                 # noqa: E402
from ecsimd_amd.curves import NAMED                                          # noqa: E402
from helpers import CURVE_PARAMS, SEED                                        # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log2n
eng = Engine(0)
mads, ms = eng.peak_mad32(8192, reps=5)
peak = mads / (ms * 1e-3) / 1e12
print(f"2^{log2n} lanes per call; measured v_mad_u64_u32 peak {peak:.2f} T mad32/s")
# algorithmic field multiplications per scalar multiplication (a multiplication = 136 mad32): the ladder 4 088; the window loop
# 63 x (25 + 19) + the table's 66 + 11 at the ends = 2 849; both + the shared inversion's share (3 per lane + 1 / 128 of an inversion) -- stated per line
LADDER_MULTS, WINDOW_MULTS = 4088, 63 * 44 + 66 + 11


def timed(fn):
    fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))


def rates(cid, label, base=None):
    k = eng.fill_random(n, SEED, 1); s = eng.fill_random(n, SEED, 2)
    bx, by = eng.scalar_mult_base(base if base is not None else cid, s, flags=OUT_AFFINE)
    res = {}
    for fl, what, mults in ((OUT_AFFINE, "ladder + shared inversion", LADDER_MULTS), (OUT_AFFINE | ALG_WINDOWED, "window loop (ALG_WINDOWED)", WINDOW_MULTS),
                            (OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME, "constant-time window loop", WINDOW_MULTS)):
        out = [eng.empty(n), eng.empty(n), None]                # (x, y, no z: OUT_AFFINE)
        t = timed(lambda: eng.scalar_mult(cid, k, bx, by, flags=fl, out=out))
        r = n / (t * 1e-3)
        print(f"{label:34s} {what:30s} {t:9.2f} ms  {r / 1e6:8.2f} M/s  {r * mults * 136 / 1e12 / peak:6.3f} of the measured multiply peak ({mults} field multiplications x 136 mad32)")
        res[fl] = [eng.to_numpy(o) for o in out[:2]]
    a, b, c_ = res[OUT_AFFINE], res[OUT_AFFINE | ALG_WINDOWED], res[OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME]
    differ = int(((a[0] != b[0]).any(axis=1) | (a[1] != b[1]).any(axis=1) | (a[0] != c_[0]).any(axis=1) | (a[1] != c_[1]).any(axis=1)).sum())
    print(f"{label:34s} lanes where the three differ: {differ} of {n}")
    return b


for name in ("brainpoolP256r1", "sm2", "frp256v1"):
    c = NAMED[name]
    cid = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"])
    rates(cid, name)
for cv, nm in ((P256, "P-256"), (SECP256K1, "secp256k1")):
    c = CURVE_PARAMS[cv]
    gid = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"], generic_kernels=True)
    g = rates(gid, f"{nm} through the generic kernels", base=cv)
    b = rates(cv, f"{nm} built-in")
    assert all(np.array_equal(u, v) for u, v in zip(g, b)), "generic window loop != built-in window loop"

# k G on a registered curve: the 4-bit comb, the constant-time 5-bit comb (every entry of a window read: 51 additions) and the signed 7-bit comb (ALG_WINDOWED_SIGNED: 36 additions instead of 63)
from ecsimd_amd import ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG     # noqa: E402
for name in ("brainpoolP256r1", "sm2", "frp256v1"):
    c = NAMED[name]
    cid = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"])
    k = eng.fill_random(n, SEED, 31)
    res = {}
    for fl, what in ((ALG_WINDOWED, "4-bit comb"), (ALG_WINDOWED | ALG_CONSTANT_TIME, "5-bit comb, constant time"), (ALG_WINDOWED_SIGNED, "signed 7-bit comb"), (ALG_WINDOWED_BIG, "20-bit comb, device memory")):
        out = [eng.empty(n), eng.empty(n), None]
        t = timed(lambda: eng.scalar_mult_base(cid, k, flags=OUT_AFFINE | fl, out=out))
        print(f"{name:34s} k G, {what:28s} {t:9.2f} ms  {n / (t * 1e-3) / 1e6:8.2f} M/s")
        res[fl] = [eng.to_numpy(o) for o in out[:2]]
    a = res[ALG_WINDOWED]
    differ = sum(int(((a[0] != b[0]).any(axis=1) | (a[1] != b[1]).any(axis=1)).sum()) for b in (res[ALG_WINDOWED | ALG_CONSTANT_TIME], res[ALG_WINDOWED_SIGNED], res[ALG_WINDOWED_BIG]))
    print(f"{name:34s} lanes where the four combs differ: {differ} of {n}")

# ECDSA verification on a registered curve: u1 G from the signed comb, u2 Q from the window loop (before: the 4-bit comb and a ladder pass)
c = NAMED["brainpoolP256r1"]
cid = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"])
m = min(n, 1 << 22)
e = eng.fill_random(m, SEED, 11); d = eng.fill_random(m, SEED, 12, clear_top_bits=2); kk = eng.fill_random(m, SEED, 13, clear_top_bits=2)
r_, s_, ok = eng.ecdsa_sign(cid, e, d, kk)
qx, qy = eng.scalar_mult_base(cid, d, flags=OUT_AFFINE)
ts = timed(lambda: eng.ecdsa_sign(cid, e, d, kk))
print(f"brainpoolP256r1 ecdsa_sign:   {ts:9.2f} ms per 2^{m.bit_length() - 1} signatures  {m / (ts * 1e-3) / 1e6:8.2f} M/s   (k G on the constant-time 5-bit comb)")
t = timed(lambda: eng.ecdsa_verify(cid, e, r_, s_, qx, qy))
good = int(eng.to_numpy(eng.ecdsa_verify(cid, e, r_, s_, qx, qy)).sum()), int(eng.to_numpy(ok).sum())
print(f"brainpoolP256r1 ecdsa_verify: {t:9.2f} ms per 2^{m.bit_length() - 1} signatures  {m / (t * 1e-3) / 1e6:8.2f} M/s   ({good[0]} of {good[1]} signed ones accepted)")
