#!/usr/bin/env python3
"""Does a short kernel run at the clock a long one gets?  Times ZDAU / TRPLU / ADD_Z2_1 / mgry_mul (a) one launch at a
time with a host synchronisation between launches (tools/point_kernels.py, tools/bench_kernels.py) and (b) as a train of
back-to-back launches right after 0.3 s of ladder work.  Usage: clock_probe.py [log2 lanes] [launches in the train]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecsimd_amd import Engine, P256

SEED = 0x5EEDEC51D0000001
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
train = int(sys.argv[2]) if len(sys.argv) > 2 else 200
e = Engine(0)
n = 1 << log2n
cv = P256
s = e.fill_random(n, SEED, 2)
k = e.fill_random(1 << 22, SEED, 1)
bx, by = e.scalar_mult_base(cv, s, flags=2)
P = e.from_affine(cv, bx, by)
T = e.trplu(cv, P)
a, b = e.fill_random(n, SEED, 3), e.fill_random(n, SEED, 4)


def ev():
    return torch.cuda.Event(enable_timing=True)


def one_at_a_time(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0, t1 = ev(), ev()
        t0.record(); fn(); t1.record(); torch.cuda.synchronize(); ts.append(t0.elapsed_time(t1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def in_a_train(fn):
    for _ in range(4):
        e.scalar_mult_base(cv, k)                 # ~0.35 s of VALU-bound work: clocks are up
    t0, t1 = ev(), ev()
    t0.record()
    for _ in range(train):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / train


outs = [e.empty(n) for _ in range(3)]
cases = {
    "zdau": lambda: e._call("zdau", __import__("ctypes").c_int(cv), *[e._ptr(t) for t in T], *[e._ptr(t) for t in P], *[e._ptr(t) for t in outs], __import__("ctypes").c_size_t(n)),
    "add_z2_1": lambda: e._call("add_z2_1", __import__("ctypes").c_int(cv), *[e._ptr(t) for t in T], e._ptr(P[0]), e._ptr(P[1]), *[e._ptr(t) for t in outs], __import__("ctypes").c_size_t(n)),
    "mgry_mul": lambda: e._call("mgry_mul", __import__("ctypes").c_int(cv), e._ptr(a), e._ptr(b), e._ptr(outs[0]), __import__("ctypes").c_size_t(n)),
}
for name, fn in cases.items():
    x, y = one_at_a_time(fn), in_a_train(fn)
    print(f"{name:10s} 2^{log2n}: one at a time {x:8.1f} us   in a train of {train} {y:8.1f} us   ({x / y:.2f}x)")
