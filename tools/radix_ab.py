"""A/B of the ladder's loop representation (round 4): radix 2^32 canonical words against nine signed 29-bit limbs.
  * ecsimd_hip_zdau_repeat -- ZDAU 254 times in registers, both radices: kernel time, cycles per iteration at the measured clock model
    (256 CUs x 4 SIMDs, 2.4 GHz nominal), and that both return the same bits;
  * the ladder itself (2^LOG2 lanes): default (radix 29) against ECSIMD_HIP_LADDER_RADIX32, same bits, M scalar mults/s.
Usage: python tools/radix_ab.py [log2 lanes = 22] [curves = p256,secp256k1]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ecsimd_amd import Engine, P256, SECP256K1, LADDER_RADIX32
from helpers import SEED

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 22
names = (sys.argv[2] if len(sys.argv) > 2 else "p256,secp256k1").split(",")
e = Engine(0)
n = 1 << log2


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(reps):
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


for nm in names:
    cv = {"p256": P256, "secp256k1": SECP256K1}[nm]
    k = e.fill_random(n, SEED, 1); s = e.fill_random(n, SEED, 2)
    bx, by = e.scalar_mult_base(cv, s, flags=2 | 32)
    P = e.from_affine(cv, bx, by)
    Pu = tuple(t.clone() for t in P)
    R = e.trplu(cv, Pu)                                   # Pu is rewritten in place with R's Z (co-Z pair)
    iters = 254
    res = {}
    for radix in (32, 29):
        out = e.zdau_repeat(cv, R, (Pu[0], Pu[1]), iters, 0x5a5a5a5a5a5a5a5a, radix)
        ms = timed(lambda: e.zdau_repeat(cv, R, (Pu[0], Pu[1]), iters, 0x5a5a5a5a5a5a5a5a, radix))
        waves_per_simd = n / 64 / (256 * 4)
        cyc = ms * 1e-3 * 2.4e9 / (waves_per_simd * iters)
        res[radix] = (out, ms, cyc)
        print(f"{nm}: zdau_repeat radix {radix}: {ms:8.3f} ms for 2^{log2} x {iters} iterations = {cyc:8.0f} cycles per wave-iteration at 2.4 GHz")
    same = all(torch.equal(a, b) for a, b in zip(res[29][0], res[32][0]))
    print(f"{nm}: zdau_repeat radix 29 == radix 32: {same}; time ratio 29/32 = {res[29][1] / res[32][1]:.4f}")
    outs = {}
    for label, fl in (("radix 29 (default)", 1), ("radix 32 (LADDER_RADIX32)", 1 | LADDER_RADIX32)):
        out = [e.empty(n) for _ in range(3)]
        ms = timed(lambda: e.scalar_mult(cv, k, P[0], P[1], flags=fl, out=out))
        outs[label] = (out, ms)
        print(f"{nm}: ladder {label}: {ms:8.3f} ms = {n / ms / 1e3:7.2f} M scalar mults/s")
    (o1, m1), (o2, m2) = outs.values()
    print(f"{nm}: ladder outputs identical: {all(torch.equal(a, b) for a, b in zip(o1, o2))}; speed-up {m2 / m1:.4f}")
