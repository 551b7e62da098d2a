"""Quick GPU parity probe used during development (the real tests live in tests/)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ecsimd_amd import Engine, P256, SECP256K1
from oracle.loader import Oracle, to_int, ints_to_arr, from_hex, from_int

o = Oracle(); e = Engine(0)
rng = np.random.default_rng(7)
def rnd(n, w=4): return rng.integers(0, 2**64, size=(n, w), dtype=np.uint64)
def chk(name, got, exp):
    got = Engine.to_numpy(got) if hasattr(got, "is_cuda") else got
    ok = np.array_equal(got, exp); bad = 0 if ok else int((got != exp).any(axis=-1).sum()) if got.ndim > 1 else int((got != exp).sum())
    print(f"{'OK ' if ok else 'FAIL'} {name}" + ("" if ok else f"  ({bad} mismatching rows; first got={got[0]} exp={exp[0]})"))
    return ok
allok = True
n = 1000
a, b = rnd(n), rnd(n); da, db = e.to_device(a), e.to_device(b)
s, f = e.add(da, db); es, ef = o.add(a, b); allok &= chk("add", s, es) & chk("add.carry", f, ef)
s, f = e.sub(da, db); es, ef = o.sub(a, b); allok &= chk("sub", s, es) & chk("sub.borrow", f, ef)
allok &= chk("mul", e.mul(da, db), o.mul(a, b)); allok &= chk("square", e.square(da), o.square(a))
s, f = e.shift_left_one(da); es, ef = o.shift_left_one(a); allok &= chk("shl1", s, es) & chk("shl1.c", f, ef)
for cv in (P256, SECP256K1):
    p = to_int(o.constants(cv)["p"])
    a = ints_to_arr([to_int(x) % p for x in rnd(n)]); b = ints_to_arr([to_int(x) % p for x in rnd(n)])
    a[0] = from_int(0); a[1] = from_int(p - 1); b[1] = from_int(p - 1); b[2] = from_int(0); a[3] = from_int(1)
    da, db = e.to_device(a), e.to_device(b)
    for nm in ("mod_add", "mod_sub", "mgry_mul"):
        allok &= chk(f"{cv}.{nm}", getattr(e, nm)(cv, da, db), getattr(o, nm)(cv, a, b))
    for nm in ("mgry_sqr", "mgry_from_classical", "mgry_to_classical", "gfp_opposite", "gfp_inverse"):
        allok &= chk(f"{cv}.{nm}", getattr(e, nm)(cv, da), getattr(o, nm)(cv, a))
    allok &= chk(f"{cv}.shl3", e.mod_shift_left(cv, da, 3), o.mod_shift_left(cv, a, 3))
    t8 = rnd(n, 8); t8[:, 7] >>= np.uint64(8)
    allok &= chk(f"{cv}.reduce", e.mgry_reduce(cv, e.to_device(t8)), o.mgry_reduce(cv, t8))
    c = o.constants(cv)
    m = 256
    k = rnd(m); k[0] = from_int(5); k[1] = from_hex("0bc1b1f28709decb543d9677d2cc9942348f6b984deff409430740942ff38827"); k[2] = from_hex("0a891cecc2bf13b0aca744434a9c9f4bd7bf5c8ed86e2f76e7df72bad813bd80")
    k[3] = from_int(0); k[4] = from_int(1); k[5] = from_int(2); k[6] = from_int(2**256 - 1)
    gx = np.tile(c["gx"], (m, 1)); gy = np.tile(c["gy"], (m, 1))
    t = time.time(); exp = o.scalar_mult(cv, k, gx, gy, threads=8); print("oracle %.2fs" % (time.time() - t))
    got = e.scalar_mult(cv, e.to_device(k), e.to_device(gx), e.to_device(gy)); torch.cuda.synchronize()
    for nm, g, x in zip("XYZ", got, exp): allok &= chk(f"{cv}.scalar_mult.{nm}", g, x)
    got = e.scalar_mult_base(cv, e.to_device(k))
    for nm, g, x in zip("XYZ", got, exp): allok &= chk(f"{cv}.scalar_mult_base.{nm}", g, x)
    ax, ay = o.to_affine(cv, exp)
    got = e.scalar_mult(cv, e.to_device(k), e.to_device(gx), e.to_device(gy), flags=2)
    # k=0 gives Z=0 -> affine meaningless but deterministic (0^(p-2) = 0): still compare
    allok &= chk(f"{cv}.scalar_mult.affine.x", got[0], ax) & chk(f"{cv}.scalar_mult.affine.y", got[1], ay)
    # lane-distinct bases: P_i = affine(k_i G), then k'_i * P_i
    valid = np.array([to_int(z) != 0 for z in exp[2]])
    bx, by = ax[valid], ay[valid]; k2 = rnd(len(bx))
    exp2 = o.scalar_mult(cv, k2, bx, by, threads=8)
    got2 = e.scalar_mult(cv, e.to_device(k2), e.to_device(bx), e.to_device(by))
    for nm, g, x in zip("XYZ", got2, exp2): allok &= chk(f"{cv}.scalar_mult(var-base).{nm}", g, x)
    # point ops
    P = o.from_affine(cv, bx, by); dP = tuple(e.to_device(v) for v in P)
    R, Pu = o.trplu(cv, P); dR = e.trplu(cv, dP)
    for nm, g, x in zip(["rx","ry","rz","px","py","pz"], list(dR) + list(dP), list(R) + list(Pu)): allok &= chk(f"{cv}.trplu.{nm}", g, x)
    R2, Qu = o.zdau(cv, R, Pu); dQ = tuple(t.clone() for t in dP); dR2 = e.zdau(cv, dR, dQ)
    for nm, g, x in zip(["rx","ry","rz","qx","qy","qz"], list(dR2) + list(dQ), list(R2) + list(Qu)): allok &= chk(f"{cv}.zdau.{nm}", g, x)
# throughput
for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    n = 1 << 20
    k = e.fill_random(n, 0x5EEDEC51D0000001, 1); x0 = e.fill_random(4, 1, 2)
    c = o.constants(cv); gx = e.to_device(np.tile(c["gx"], (n, 1))); gy = e.to_device(np.tile(c["gy"], (n, 1)))
    out = [e.empty(n) for _ in range(3)]
    e.scalar_mult(cv, k, gx, gy, out=out); torch.cuda.synchronize()
    t = time.time(); e.scalar_mult(cv, k, gx, gy, out=out); torch.cuda.synchronize(); dt = time.time() - t
    print(f"{nm}: {n} scalar mults in {dt*1e3:.1f} ms -> {n/dt/1e6:.2f} M/s")
mads, ms = e.peak_mad32(4096); print(f"peak mad32: {mads/ms/1e9:.2f} T/s")
print("ALL OK" if allok else "SOME FAILED"); sys.exit(0 if allok else 1)
