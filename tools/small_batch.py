#!/usr/bin/env python3
"""Small batches: what one call costs on the GPU against the compiled reference on the host.

For n = 4 .. 2^16 lanes, wall time of ONE synchronous call (host clock around call + stream sync, best of 7 after a warm-up):
  * scalar_mult, variable base, Jacobian out        -- what the drop-in scalar_mult_p256 adapter calls (the 254-iteration ladder);
  * scalar_mult, variable base, affine out          -- the same + the shared inversion;
  * scalar_mult_base, affine out, default flags     -- k*G: the constant-time comb up to 2^16 lanes (r4), the ladder above;
  * scalar_mult_base, affine out, ALG_WINDOWED_BIG  -- k*G for public scalars: 12 additions over the 20-bit table;
against the reference's own call shape on this host (oracle/_ref: curve_group<P256>::scalar_mult over wides of 4 lanes, then to_affine()
as benchs/curve_group.cpp:23-35 times it) on 1 thread and on all cores.  Prints the table INTEGRATION.md section 2 quotes and the
crossover batch sizes.  `python tools/small_batch.py [p256|secp256k1]`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from ecsimd_amd import Engine, CURVES, OUT_AFFINE, ALG_WINDOWED_BIG, BASE_MGRY
from helpers import SEED
from oracle.loader import Reference, reference_available

cv = CURVES[sys.argv[1] if len(sys.argv) > 1 else "p256"]
e = Engine(0)
ref = Reference() if reference_available() else None
cores = len(os.sched_getaffinity(0))


def best(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts)


def cpu(n, k, gx, gy, threads):
    if ref is None:
        return float("nan")
    ts = []
    for _ in range(3 if n <= 4096 else 1):
        t = time.perf_counter(); j = ref.scalar_mult(cv, k, gx, gy, threads=threads); ts.append(time.perf_counter() - t)
    return min(ts)


c = None
print(f"device {torch.cuda.get_device_name(0)}; host cores {cores}; reference {'oracle/_ref' if ref else 'absent'}")
print(f"{'lanes':>7} | {'var J ms':>9} {'var aff ms':>10} {'base aff ms':>11} {'base big ms':>11} | {'ref 1 thr ms':>12} {'ref all ms':>10} | GPU base/var-J faster than 1 thread, all cores")
rows = []
for lg in (2, 4, 6, 8, 10, 12, 14, 16, 17):
    n = 1 << lg
    k = e.fill_random(n, SEED, 1); s = e.fill_random(n, SEED, 2)
    bx, by = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED_BIG)
    P = e.from_affine(cv, bx, by)
    outj = [e.empty(n) for _ in range(3)]
    t_vj = best(lambda: e.scalar_mult(cv, k, P[0], P[1], flags=BASE_MGRY, out=outj))
    t_va = best(lambda: e.scalar_mult(cv, k, P[0], P[1], flags=BASE_MGRY | OUT_AFFINE, out=outj))
    t_b = best(lambda: e.scalar_mult_base(cv, k, flags=OUT_AFFINE, out=outj))
    t_big = best(lambda: e.scalar_mult_base(cv, k, flags=OUT_AFFINE | ALG_WINDOWED_BIG, out=outj))
    kn, xn, yn = (Engine.to_numpy(t) for t in (k, bx, by))
    t1 = cpu(n, kn, xn, yn, 1) if n <= (1 << 14) else float("nan")
    ta = cpu(n, kn, xn, yn, cores)
    rows.append((n, t_vj, t_va, t_b, t_big, t1, ta))
    print(f"{n:>7} | {t_vj*1e3:9.3f} {t_va*1e3:10.3f} {t_b*1e3:11.3f} {t_big*1e3:11.3f} | {t1*1e3:12.3f} {ta*1e3:10.3f} | "
          f"base {'yes' if t_b < t1 else 'no':>3} {'yes' if t_b < ta else 'no':>3}   var-J {'yes' if t_vj < t1 else 'no':>3} {'yes' if t_vj < ta else 'no':>3}", flush=True)
if ref is not None:
    per1 = min(r[5] / r[0] for r in rows if r[5] == r[5])
    pera = min(r[6] / r[0] for r in rows if r[6] == r[6])
    print(f"reference: {1 / per1:,.0f} scalar mults/s on one thread, {1 / pera:,.0f} on {cores} threads (best over the sizes above)")
    for name, col in (("variable base, Jacobian (the adapter's call)", 1), ("k*G affine, default flags", 3)):
        floor = min(r[col] for r in rows[:4])
        print(f"{name}: floor {floor*1e3:.3f} ms = {floor / per1:,.0f} lanes of one host thread, {floor / pera:,.0f} lanes of {cores} threads")
