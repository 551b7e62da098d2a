#!/usr/bin/env python3
"""SEC1 decompression (a square root per point: 253 squarings) at 2^22 points on both curves, M points/s -- the A/B of the square-root chains
(ECSIMD_HIP_LIBRARY=<other build> python tools/decode_perf.py).  The decoded points are compared with the points that were encoded."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ecsimd_amd import Engine, P256, SECP256K1, OUT_AFFINE
e = Engine(0); n = 1 << 22
for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    s = e.fill_random(n, 7, 2)
    bx, by = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | 8)
    wire = e.sec1_encode(cv, bx, by, True)
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = e.sec1_decode(cv, wire, True); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e-3)
    x, y, ok = out
    same = bool(torch.equal(x, bx) and torch.equal(y, by) and bool(ok.all()))
    print(f"sec1_decode<{nm}> compressed, 2^22 points: {n / sorted(ts[1:])[2] / 1e6:8.1f} M points/s   decoded == encoded: {same}", flush=True)
    if not same:
        sys.exit(1)
