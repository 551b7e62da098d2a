"""Development loop for the variable-base windowed path: parity (edge scalars + random, vs the ladder's affine
output and the oracle) and throughput at 2^21 / 2^22."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ecsimd_amd import Engine, P256, SECP256K1, OUT_AFFINE, ALG_WINDOWED, BASE_MGRY
from helpers import fill_random_np, SEED, CURVE_PARAMS, ints_to_arr, to_int, ec_mul
e = Engine(0)
ok = True
for cv in (P256, SECP256K1):
    c = CURVE_PARAMS[cv]; order = c["n"]
    edge = [0, 1, 2, 7, 8, 9, 15, 16, 17, 0x88, 0x80, 0x78, 2**252, 2**255, (order - 1) // 2, (order + 1) // 2, order - 2, order - 1, order, order + 1,
            order + 9, 2**256 - 1, int("8" * 64, 16), int("7" * 64, 16), int("9" * 64, 16), int("f" * 63 + "8", 16) % 2**256, int("08" * 32, 16), int("80" * 32, 16)]
    n = 4096 + 13
    k = fill_random_np(n, SEED, 51); k[:len(edge)] = ints_to_arr(edge)
    s = fill_random_np(n, SEED, 52)
    bx, by = e.scalar_mult_base(cv, e.to_device(s), flags=OUT_AFFINE | ALG_WINDOWED)
    dk = e.to_device(k)
    wx, wy = e.scalar_mult(cv, dk, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    lx, ly = e.scalar_mult(cv, dk, bx, by, flags=OUT_AFFINE)
    wxn, wyn, lxn, lyn, bxn, byn = (e.to_numpy(t) for t in (wx, wy, lx, ly, bx, by))
    bad = 0
    for i in range(len(edge)):
        exp = ec_mul(cv, edge[i] % order, (to_int(bxn[i]), to_int(byn[i])))
        exp = (0, 0) if exp is None else exp
        if (to_int(wxn[i]), to_int(wyn[i])) != exp:
            bad += 1; print("edge mismatch", cv, hex(edge[i]))
    rest = slice(len(edge), n)
    good = np.array_equal(wxn[rest], lxn[rest]) and np.array_equal(wyn[rest], lyn[rest]) and bad == 0
    P = e.from_affine(cv, bx, by)                                   # Montgomery-form base
    mx, my = e.scalar_mult(cv, dk, P[0], P[1], flags=OUT_AFFINE | ALG_WINDOWED | BASE_MGRY)
    good &= torch.equal(mx, wx) and torch.equal(my, wy)
    px, py = e.scalar_mult(cv, dk, bx, by, flags=OUT_AFFINE | ALG_WINDOWED | 16)      # ALG_NO_ENDOMORPHISM: the plain loop
    good &= torch.equal(px, wx) and torch.equal(py, wy)
    print("varwin parity", cv, "OK" if good else "FAIL"); ok &= good
if not ok:
    sys.exit(1)
for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    for lg in (21, 22):
        n = 1 << lg
        k = e.fill_random(n, SEED, 1); s = e.fill_random(n, SEED, 2)
        bx, by = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED)
        out = [e.empty(n) for _ in range(2)]
        for flags, label in ((OUT_AFFINE | ALG_WINDOWED, "windowed"), (OUT_AFFINE, "ladder + to_affine")):
            e.scalar_mult(cv, k, bx, by, flags=flags, out=out + [None]); torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                t = time.time(); e.scalar_mult(cv, k, bx, by, flags=flags, out=out + [None]); torch.cuda.synchronize(); ts.append(time.time() - t)
            print(f"{nm} 2^{lg} variable base, affine out, {label}: {n/min(ts)/1e6:.2f} M/s ({min(ts)*1e3:.2f} ms)", flush=True)
