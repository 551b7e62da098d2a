#!/usr/bin/env python3
"""Kernel time of the stand-alone ZDAU (config 2) at 2^20 .. 2^24 points, HIP events around 20 in-place launches (the result feeds the
next launch: timing is data-independent).  ECSIMD_HIP_LIBRARY selects the build."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from ecsimd_amd import Engine, P256, SECP256K1
e = Engine(0)
for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    for log2 in (20, 22, 24):
        n = 1 << log2
        s = e.fill_random(n, 7, 2); bx, by = e.scalar_mult_base(cv, s, flags=2); P = e.from_affine(cv, bx, by)
        Q = tuple(t.clone() for t in P); T = e.trplu(cv, Q)          # (3P, P') share Z
        r = [e.empty(n) for _ in range(3)]
        def launch():
            e._call("zdau", C.c_int(cv), *[e._ptr(t) for t in T], *[e._ptr(t) for t in Q], *[e._ptr(t) for t in r], C.c_size_t(n))
        launch(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): launch()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 20
        print(f"zdau<{nm}> 2^{log2}: {us:9.1f} us  {n / us / 1e3:8.2f} G points/s", flush=True)
        del P, Q, T, r, s, bx, by
