"""Development loop: parity spot-check (1024 lanes, both curves) + ladder throughput at 2^21."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ecsimd_amd import Engine, P256, SECP256K1
from oracle.loader import Oracle
from helpers import fill_random_np, SEED
e = Engine(0); o = Oracle()
ok = True
for cv in (P256, SECP256K1):
    n = 1024
    k = fill_random_np(n, SEED, 1); s = fill_random_np(n, SEED, 2)
    bx, by = e.scalar_mult_base(cv, e.to_device(s), flags=2)
    got = e.scalar_mult(cv, e.to_device(k), bx, by)
    exp = o.scalar_mult(cv, k, e.to_numpy(bx), e.to_numpy(by), threads=16)
    good = all(np.array_equal(e.to_numpy(g), x) for g, x in zip(got, exp))
    ga = e.scalar_mult(cv, e.to_device(k), bx, by, flags=2); xa = o.to_affine(cv, exp)
    good &= all(np.array_equal(e.to_numpy(g), x) for g, x in zip(ga, xa))
    print("parity", cv, "OK" if good else "FAIL"); ok &= good
for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    n = 1 << 21
    k = e.fill_random(n, SEED, 1); s = e.fill_random(n, SEED, 2)
    bx, by = e.scalar_mult_base(cv, s, flags=2); P = e.from_affine(cv, bx, by)
    out = [e.empty(n) for _ in range(3)]
    e.scalar_mult(cv, k, P[0], P[1], flags=1, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.time(); e.scalar_mult(cv, k, P[0], P[1], flags=1, out=out); torch.cuda.synchronize(); ts.append(time.time() - t)
    dt = min(ts)
    print(f"{nm}: ladder {n/dt/1e6:.2f} M/s  ({dt*1e3:.2f} ms for 2^21)")
    ox, oy = e.scalar_mult_base(cv, k, flags=2 | 4); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.time(); e.scalar_mult_base(cv, k, flags=2 | 4); torch.cuda.synchronize(); ts.append(time.time() - t)
    print(f"{nm}: fixed-base windowed (affine out) {n/min(ts)/1e6:.2f} M/s")
    J = e.scalar_mult(cv, k, P[0], P[1], flags=1)
    ts = []
    for _ in range(5):
        t = time.time(); e.to_affine(cv, J); torch.cuda.synchronize(); ts.append(time.time() - t)
    print(f"{nm}: to_affine (simultaneous inversion) {n/min(ts)/1e6:.2f} M/s")
mads, ms = e.peak_mad32(8192, reps=5); print(f"peak mad32: {mads/ms/1e9:.2f} T/s")
sys.exit(0 if ok else 1)
