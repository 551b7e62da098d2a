"""Throughput and latency of the ladder and the windowed variable-base path against the batch size (P-256)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ecsimd_amd import Engine, P256, OUT_AFFINE, ALG_WINDOWED, BASE_MGRY
from helpers import SEED
e = Engine(0)
print(f"{'batch':>9}  {'ladder ms':>10} {'M/s':>8}   {'windowed ms':>11} {'M/s':>8}")
for lg in (2, 6, 10, 12, 14, 15, 16, 17, 18, 19, 20, 22):
    n = 1 << lg
    k = e.fill_random(n, SEED, 1); s = e.fill_random(n, SEED, 2)
    bx, by = e.scalar_mult_base(P256, s, flags=OUT_AFFINE | ALG_WINDOWED)
    P = e.from_affine(P256, bx, by)
    outj = [e.empty(n) for _ in range(3)]
    res = []
    for fn in (lambda: e.scalar_mult(P256, k, P[0], P[1], flags=BASE_MGRY, out=outj),
               lambda: e.scalar_mult(P256, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED, out=outj)):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        res.append(min(ts))
    print(f"2^{lg:<2} {n:>6}  {res[0]*1e3:10.3f} {n/res[0]/1e6:8.2f}   {res[1]*1e3:11.3f} {n/res[1]/1e6:8.2f}", flush=True)
