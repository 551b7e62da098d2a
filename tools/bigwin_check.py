import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from ecsimd_amd import Engine, P256, SECP256K1, OUT_AFFINE, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG
from helpers import SEED, CURVE_PARAMS, ints_to_arr
e = Engine(0)
for cv in (P256, SECP256K1):
    order = CURVE_PARAMS[cv]["n"]
    n = (1 << 22) + 13
    k = e.fill_random(n, SEED, 5)
    edge = [0, 1, 2, 0x7fff, 0x8000, 0x8001, 0xffff, 0x10000, 0x18000, 0x7ffff, 0x80000, 0x80001, 0xfffff, 0x100000, int("80000" * 12, 16), int("7ffff" * 12, 16), order - 1, order, order + 1, 2**256 - 1, 2**255, int("8000" * 16, 16), int("7fff" * 16, 16), int("8001" * 16, 16), 2**256 - order]
    k[:len(edge)] = e.to_device(ints_to_arr(edge))
    t = time.time(); bx, by = e.scalar_mult_base(cv, k, flags=OUT_AFFINE | ALG_WINDOWED_BIG); torch.cuda.synchronize(); print("first call (table build)", round(time.time() - t, 3), "s")
    sx, sy = e.scalar_mult_base(cv, k, flags=OUT_AFFINE | ALG_WINDOWED_SIGNED)
    print("parity", cv, bool(torch.equal(bx, sx) and torch.equal(by, sy)))
    out = [e.empty(n) for _ in range(3)]
    for fl, nm in ((ALG_WINDOWED_BIG, "20-bit windows, table in device memory"), (ALG_WINDOWED_SIGNED, "7-bit windows, table in LDS")):
        ts = []
        for _ in range(6):
            t = time.time(); e.scalar_mult_base(cv, k, flags=OUT_AFFINE | fl, out=out); torch.cuda.synchronize(); ts.append(time.time() - t)
        print(f"{nm}: {n/min(ts)/1e6:.1f} M/s ({min(ts)*1e3:.2f} ms)")
