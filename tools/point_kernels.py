#!/usr/bin/env python3
"""Driver for rocprofv3 over the BASELINE configs[1] kernels (point add / double, batch 2^20) and their neighbours:
k_trplu, k_zdau, k_add_z2_1, k_dblu, k_zaddu, k_to_affine_batched, k_inverse_batched -- each launched `reps` times at
2^log2n lanes on both curves, events-timed as well (stderr).  tools/profile_points.sh wraps it in the kernel-trace and
PMC passes; tools/summarize_profiles.py condenses the result into profiles/<round>/point/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecsimd_amd import Engine, P256, SECP256K1

SEED = 0x5EEDEC51D0000001
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
e = Engine(0)
n = 1 << log2n


def timed(name, fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"{name:34s} 2^{log2n}: {ts[len(ts) // 2] * 1e3:9.1f} us  {n / ts[len(ts) // 2] / 1e3:10.1f} M/s", file=sys.stderr)


for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    s = e.fill_random(n, SEED, 2)
    bx, by = e.scalar_mult_base(cv, s, flags=2 | 32)
    P = e.from_affine(cv, bx, by)
    Q = tuple(t.clone() for t in P)
    T = e.trplu(cv, Q)                                    # T = 3P, Q = P re-expressed co-Z
    # in-place kernels keep rewriting the same buffers: the arithmetic stays well-defined (co-Z is preserved)
    timed(f"trplu<{nm}>", lambda: e.trplu(cv, Q))
    Q = tuple(t.clone() for t in P); T = e.trplu(cv, Q)
    timed(f"zdau<{nm}>", lambda: e.zdau(cv, T, Q))
    timed(f"add_z2_1<{nm}>", lambda: e.add_z2_1(cv, T, (P[0], P[1])))
    Q = tuple(t.clone() for t in P)
    timed(f"dblu<{nm}>", lambda: e.dblu(cv, Q))
    D = e.dblu(cv, Q)
    timed(f"zaddu<{nm}>", lambda: e.zaddu(cv, Q, D))
    timed(f"to_affine (batched)<{nm}>", lambda: e.to_affine(cv, T))
    timed(f"gfp_inverse (batched)<{nm}>", lambda: e.gfp_inverse(cv, T[2]))
    a = e.fill_random(n, SEED, 11, clear_top_bits=1)
    timed(f"mgry_sqr<{nm}>", lambda: e.mgry_sqr(cv, a))
    timed(f"mgry_mul<{nm}>", lambda: e.mgry_mul(cv, a, bx))
