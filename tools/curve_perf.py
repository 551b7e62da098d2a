#!/usr/bin/env python3
"""Round 5: what the ladder costs on a curve registered at run time (dense 9-limb prime in SGPRs) beside the built-in special-form kernels.
tools/curve_perf.py [log2 lanes]: variable-base ladder, Jacobian out, HIP-event time of 5 launches after a warm-up; the three loops of a registered curve
(29-bit limbs, canonical words, reference squaring), P-256 / secp256k1 through the generic kernels, and the built-in kernels on the same box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ecsimd_amd import Engine, P256, SECP256K1                     # noqa: E402
from ecsimd_amd.engine import register_curve                      # noqa: E402
from oracle.loader import REF_CURVES                              # noqa: E402  (parameters only: nothing of the oracle runs here)
from helpers import CURVE_PARAMS, SEED                             # noqa: E402

LADDER_RADIX32, REF_SQUARE_COMPAT, OUT_AFFINE = 256, 64, 2
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log2n
eng = Engine(0)
mads, ms = eng.peak_mad32(8192, reps=5)
peak = mads / (ms * 1e-3) / 1e12
print(f"2^{log2n} lanes per launch; measured v_mad_u64_u32 peak {peak:.2f} T mad32/s")


def rate(cid, flags, label, base=None):
    k = eng.fill_random(n, SEED, 1); s = eng.fill_random(n, SEED, 2)
    bx, by = eng.scalar_mult_base(base if base is not None else cid, s, flags=OUT_AFFINE)
    out = [eng.empty(n) for _ in range(3)]
    eng.scalar_mult(cid, k, bx, by, flags=flags, out=out)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record(); eng.scalar_mult(cid, k, bx, by, flags=flags, out=out); b.record()
    torch.cuda.synchronize()
    t = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    r = n / (t * 1e-3)
    print(f"{label:78s} {t:9.2f} ms  {r / 1e6:8.2f} M/s  {r * 555968 / 1e12 / peak:6.3f} of the measured multiply peak (algorithmic 555 968 mad32 per scalar mult)")
    return [eng.to_numpy(o[:4096]) for o in out]


for name in ("brainpoolP256r1", "sm2", "frp256v1"):
    c = REF_CURVES[name]
    cid = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"])
    a = rate(cid, 0, f"{name}: registered curve, 29-bit limbs, dense reduction (default)")
    if name == "brainpoolP256r1":
        b = rate(cid, LADDER_RADIX32, f"{name}: registered curve, 8 x 32-bit canonical words (LADDER_RADIX32)")
        assert all(np.array_equal(u, v) for u, v in zip(a, b))
        rate(cid, REF_SQUARE_COMPAT, f"{name}: registered curve, reference squaring (REF_SQUARE_COMPAT)")
for cv, nm in ((P256, "P-256"), (SECP256K1, "secp256k1")):
    c = CURVE_PARAMS[cv]
    gid = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"], generic_kernels=True)
    a = rate(cv, 0, f"{nm}: built-in special-form kernel, 29-bit limbs, sparse reduction")
    b = rate(gid, 0, f"{nm}: the same curve through the generic kernels (dense reduction)", base=cv)
    assert all(np.array_equal(u, v) for u, v in zip(a, b)), "generic != special"
    rate(cv, LADDER_RADIX32, f"{nm}: built-in, 8 x 32-bit canonical words (LADDER_RADIX32)")
print("outputs of the compared pairs are identical on the first 4096 lanes")

# small batches of k G on a registered curve: the default route (the constant-time comb up to 2^16 lanes) against the ladder launch it replaces
import time
c = REF_CURVES["brainpoolP256r1"]
cid = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"])
print("brainpoolP256r1, scalar_mult_base(OUT_AFFINE): wall time of ONE synchronous call (best of 7)")
for lg in (2, 8, 12, 16):
    m = 1 << lg
    k = eng.fill_random(m, SEED, 5)
    out = [eng.empty(m) for _ in range(3)]
    def best(fl):
        eng.scalar_mult_base(cid, k, flags=fl, out=out); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t = time.perf_counter(); eng.scalar_mult_base(cid, k, flags=fl, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        return min(ts)
    ladder = best(OUT_AFFINE | LADDER_RADIX32)                                          # an explicit ladder flag keeps the ladder (on canonical words)
    gx = eng.to_device(np.tile(np.array([(c["gx"] >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64), (m, 1)))
    gy = eng.to_device(np.tile(np.array([(c["gy"] >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64), (m, 1)))
    t0 = best(OUT_AFFINE)
    eng.scalar_mult(cid, k, gx, gy, flags=OUT_AFFINE, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t = time.perf_counter(); eng.scalar_mult(cid, k, gx, gy, flags=OUT_AFFINE, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"  2^{lg:<2d} lanes: default route {1e3 * t0:7.3f} ms; the 29-bit ladder on G as a variable base {1e3 * min(ts):7.3f} ms; LADDER_RADIX32 {1e3 * ladder:7.3f} ms")
