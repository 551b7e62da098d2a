#!/usr/bin/env python3
"""to_affine (simultaneous inversion) and the fixed-base big-window path at 2^20 / 2^22 / 2^24 points.
The round-2 sweep of how many elements share one inversion (32 / 64 / 128 / 256 per lane, 2^17 / 2^18 / 2^19 lanes) was run with
this script against builds of k_affine.inc with those two numbers overridden; 128 elements and 2^17 lanes won at every size
(to_affine 10.9 -> 13.5 G points/s at 2^24) and are what BATCH_INVERSION_MAX / the launchers now hold."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecsimd_amd import Engine, P256, OUT_AFFINE, ALG_WINDOWED_BIG
e = Engine(0)
SEED = 0x5EEDEC51D0000001
for log2n in (20, 22, 24):
    n = 1 << log2n
    k = e.fill_random(n, SEED, 1); s = e.fill_random(n, SEED, 2)
    bx, by = e.scalar_mult_base(P256, s, flags=OUT_AFFINE | ALG_WINDOWED_BIG)
    J = e.scalar_mult(P256, k, bx, by)

    def t(fn, reps=5):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]
    ms = t(lambda: e.to_affine(P256, J))
    out = [e.empty(n) for _ in range(3)]
    msb = t(lambda: e.scalar_mult_base(P256, k, flags=OUT_AFFINE | ALG_WINDOWED_BIG, out=out))
    print(f"2^{log2n}: to_affine {ms*1e3:8.1f} us = {n/ms/1e6:7.2f} G points/s   fixed-base-big {n/msb/1e3:8.1f} M/s")
