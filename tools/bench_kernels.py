#!/usr/bin/env python3
"""Secondary kernels: throughput and roofline fraction on one MI355X (the headline lives in bench.py).
Writes one JSON object (stdout) -- committed as profiles/rNN/secondary_kernels.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecsimd_amd import Engine, P256, SECP256K1

SEED = 0x5EEDEC51D0000001
e = Engine(0)


def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


mads, ms = e.peak_mad32(8192, reps=5)
peak_mad = mads / (ms * 1e-3)
out = {"device": torch.cuda.get_device_name(0), "peak_mad32_per_s": peak_mad, "hbm_peak_GBps": 8000.0, "kernels": []}


def row(name, n, t, mad32_per_unit, bytes_per_unit, unit, executed_mad32=None):
    """mad32_per_unit: SURVEY.md 8(d)'s algorithmic count (136 per field multiplication OR squaring).  executed_mad32: the multiply-adds the kernel
    really issues, where that differs much (a squaring on 29-bit limbs is 81 / 63 of them): the algorithmic fraction can then exceed 1."""
    r = {"kernel": name, "n": n, "seconds": t, "rate": n / t, "unit": unit + "/s",
         "valu": {"algorithmic_mad32_per_unit": mad32_per_unit, "achieved_Tmad32_s": n / t * mad32_per_unit / 1e12, "frac_of_measured_peak": n / t * mad32_per_unit / peak_mad},
         "hbm": {"algorithmic_bytes_per_unit": bytes_per_unit, "achieved_GBps": n / t * bytes_per_unit / 1e9, "frac_of_8TBps": n / t * bytes_per_unit / 8e12}}
    r["bound"] = "hbm" if r["hbm"]["frac_of_8TBps"] > r["valu"]["frac_of_measured_peak"] else "valu"
    note = ""
    if executed_mad32 is not None:
        r["valu"]["executed_mad32_per_unit"] = executed_mad32
        r["valu"]["executed_frac_of_measured_peak"] = n / t * executed_mad32 / peak_mad
        note = f"  ({r['valu']['executed_frac_of_measured_peak']:.2f} in multiply-adds really issued: squarings are counted as products)"
    out["kernels"].append(r)
    print(f"{name:44s} n=2^{n.bit_length()-1:<2d} {n/t/1e6:10.1f} M {unit}/s   valu {r['valu']['frac_of_measured_peak']:.2f}  hbm {r['hbm']['frac_of_8TBps']:.3f}{note}", file=sys.stderr)


n = 1 << 24
a = e.fill_random(n, SEED, 11, clear_top_bits=1); b = e.fill_random(n, SEED, 12, clear_top_bits=1)
by = e.to_bytes_be(a)
row("to_bytes_be (HBM-bound codec)", n, timeit(lambda: e.to_bytes_be(a)), 0, 64, "elements")
row("from_bytes_be", n, timeit(lambda: e.from_bytes_be(by)), 0, 64, "elements")
for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    row(f"mod_add<{nm}>", n, timeit(lambda: e.mod_add(cv, a, b)), 0, 96, "elements")
    inv_m = (255 + 12) if cv == 0 else (255 + 15)
    share = min(128, max(1, n >> 17))                 # elements that share one inversion (k_affine.inc BATCH_INVERSION_MAX)
    row(f"gfp_inverse<{nm}> (simultaneous inversion, 3 + {inv_m}/{share} field mults)", n, timeit(lambda: e.gfp_inverse(cv, a)), int((3 + inv_m / share) * 136), 128, "elements")
    row(f"mgry_mul<{nm}> element-wise", n, timeit(lambda: e.mgry_mul(cv, a, b)), 136, 96, "field mults")
    row(f"mgry_sqr<{nm}> element-wise", n, timeit(lambda: e.mgry_sqr(cv, a)), 136, 64, "field mults")
del by
n = 1 << 20                                                   # BASELINE configs[1]: point add + double, batch 2^20
for cv, nm in ((P256, "p256"), (SECP256K1, "secp256k1")):
    s = e.fill_random(n, SEED, 2)
    bx, byy = e.scalar_mult_base(cv, s, flags=2)
    P = e.from_affine(cv, bx, byy)
    def trplu():
        Q = tuple(t.clone() for t in P); return e.trplu(cv, Q), Q
    T, Pu = trplu()
    clone_t = timeit(lambda: tuple(t.clone() for t in P))
    row(f"TRPLU<{nm}> (DBLU + ZADDU, config 2)", n, timeit(lambda: trplu()) - clone_t, 13 * 136, 64 + 192 + 96, "points")
    def zdau():
        Q = tuple(t.clone() for t in Pu); return e.zdau(cv, T, Q)
    row(f"ZDAU<{nm}> (config 2)", n, timeit(zdau) - clone_t, 16 * 136, 160 + 192, "points")
    row(f"ADD_Z2_1<{nm}>", n, timeit(lambda: e.add_z2_1(cv, T, (P[0], P[1]))), 11 * 136, 160 + 96, "points")
    n2 = 1 << 22
    k = e.fill_random(n2, SEED, 1); s2 = e.fill_random(n2, SEED, 2)
    b2x, b2y = e.scalar_mult_base(cv, s2, flags=2); P2 = e.from_affine(cv, b2x, b2y)
    outj = [e.empty(n2) for _ in range(3)]
    row(f"scalar_mult<{nm}> ladder, Jacobian out", n2, timeit(lambda: e.scalar_mult(cv, k, P2[0], P2[1], flags=1, out=outj), 5), 555968, 192, "scalar mults")
    row(f"scalar_mult<{nm}> ladder, affine out", n2, timeit(lambda: e.scalar_mult(cv, k, P2[0], P2[1], flags=3, out=outj), 5), 555968 + 19 * 136, 160, "scalar mults")
    # x only: P-256 runs the ladder without Z (8M + 6S per bit, point.cuh scalar_mult_ladder_x); secp256k1 (a = 0) the full ladder + an x-only conversion
    row(f"scalar_mult<{nm}> ladder, x coordinate only (ECDH)", n2, timeit(lambda: e.scalar_mult(cv, k, P2[0], P2[1], flags=3, out=[outj[0], None, None]), 5),
        ((6 + 7) + 254 * 14 + 20 + 7) * 136 if cv == 0 else 555968 + 17 * 136, 128, "scalar mults")
    dblm = 8 if cv == 0 else 7
    vw = int((55 + (7 * 5 + 3 + inv_m / 32) + 63 * (3 * dblm + 18) + (7 + inv_m / 32)) * 136)          # odd digits, fused double-add; the table by one inversion per lane + the walk back
    if cv == 1:                                      # secp256k1: GLV split (k_varwin.inc)
        vw = int(((dblm + 4 + 6 * 7 + 7 * 5) + 32 * (4 * dblm + 23) + 23 + 1 + (7 + inv_m / 32)) * 136)       # the table over one Z (k_varwin_table_iso), the loop on the isomorphic curve
    row(f"scalar_mult<{nm}> windowed variable base (per-element tables), affine out", n2,
        timeit(lambda: e.scalar_mult(cv, k, b2x, b2y, flags=2 | 4, out=outj), 5), vw, 160, "scalar mults")
    row(f"to_affine<{nm}> (simultaneous inversion)", n2, timeit(lambda: e.to_affine(cv, outj)), int((7 + inv_m / 32) * 136), 256, "points")
    row(f"scalar_mult_base<{nm}> windowed, affine out", n2, timeit(lambda: e.scalar_mult_base(cv, k, flags=6, out=outj)), int((64 * 11 + 7 + inv_m / 32) * 136), 96, "scalar mults")
    u1 = e.fill_random(n2, SEED, 21)
    row(f"double_scalar_mult<{nm}> u1*G + u2*Q (ECDSA-verify shape)", n2, timeit(lambda: e.double_scalar_mult(cv, u1, k, b2x, b2y), 5),
        vw + int((12 * 11 + (7 + inv_m / 32) + 6 + inv_m / 32) * 136), 160, "verifications")
    # the whole verification: + the arithmetic modulo the group order (k_gfield.hip k_ecdsa_scalars: 6 generic Montgomery products per signature and
    # one division-step inversion per 32 at this size) and the final x mod n == r; (e, r, s, Qx, Qy) in = 160 B
    dsm = vw + int((12 * 11 + (7 + inv_m / 32) + 6 + inv_m / 32) * 136)
    row(f"ecdsa_verify_rx<{nm}> (u1, u2 given)", n2, timeit(lambda: e.ecdsa_verify_rx(cv, u1, k, b2x, b2y, u1), 5), dsm, 161, "verifications")
    rr = e.fill_random(n2, SEED, 22, clear_top_bits=1); ss = e.fill_random(n2, SEED, 23, clear_top_bits=1)
    row(f"ecdsa_verify<{nm}> (e, r, s, Q -> ok: mod-n arithmetic on the device)", n2, timeit(lambda: e.ecdsa_verify(cv, u1, rr, ss, b2x, b2y), 5), dsm + 6 * 136, 161, "verifications")
    row(f"ecdsa_sign<{nm}> (e, d, k -> r, s: k G on the constant-time comb + mod-n arithmetic on the device)", n2, timeit(lambda: e.ecdsa_sign(cv, u1, rr, ss), 5),
        int((51 * 11 + 9 + 7) * 136), 160, "signatures")
    from ecsimd_amd.engine import ORDER_FIELD
    fo = ORDER_FIELD[cv]
    row(f"mgry_mul<order of {nm}> (run-time modulus, generic reduction)", n2, timeit(lambda: e.mgry_mul(fo, rr, ss)), 136, 96, "field mults")
    row(f"gfp_inverse<order of {nm}> (division steps, shared by 32)", n2, timeit(lambda: e.gfp_inverse(fo, rr)), int((3 + 88 / 32) * 136), 128, "elements")   # ~12 000 instructions of division steps ~ 88 field multiplications, shared by 32
    del u1, rr, ss
    row(f"scalar_mult_base<{nm}> 20-bit windows, odd digits (table in device memory), affine out", n2, timeit(lambda: e.scalar_mult_base(cv, k, flags=2 | 32, out=outj)), int((12 * 11 + 7 + inv_m / 32) * 136), 96, "scalar mults")
    row(f"scalar_mult_base<{nm}> signed 7-bit windows, affine out", n2, timeit(lambda: e.scalar_mult_base(cv, k, flags=2 | 8, out=outj)), int((37 * 11 + 7 + inv_m / 32) * 136), 96, "scalar mults")
    wire = e.sec1_encode(cv, b2x, b2y, True)
    row(f"sec1_decode<{nm}> compressed (decompression)", n2, timeit(lambda: e.sec1_decode(cv, wire, True)), ((253 + 7 + 4) if cv == 0 else (253 + 13 + 4)) * 136, 33 + 64, "points",
        executed_mad32=(253 * 81 + 9 * 117 + 3 * 100) if cv == 0 else (253 * 63 + 15 * 99 + 3 * 80))       # the chain on 29-bit limbs: sqr29 81 / 63, mul29 117 / 99 multiply-adds; right-hand side and check on canonical words
    wire = e.sec1_encode(cv, b2x, b2y, False)
    row(f"sec1_decode<{nm}> uncompressed (validation)", n2, timeit(lambda: e.sec1_decode(cv, wire, False)), 4 * 136, 65 + 64, "points")
    del k, s2, b2x, b2y, P2, outj, wire
# (r5) a curve registered at run time: the reference's layers on the generic kernels, its two table-driven algorithms, and the first application on top of them
from ecsimd_amd.curves import curve_id
for nm in ("brainpoolP256r1",):
    cv = curve_id(nm)
    n2 = 1 << 22
    k = e.fill_random(n2, SEED, 1, clear_top_bits=1); s2 = e.fill_random(n2, SEED, 2)
    b2x, b2y = e.scalar_mult_base(cv, s2, flags=2); P2 = e.from_affine(cv, b2x, b2y)
    outj = [e.empty(n2) for _ in range(3)]
    inv_g = 255 + 12                                                   # x^(p-2)-sized work per shared inversion on the generic words (division steps in fact: ~ 88 products)
    row(f"scalar_mult<{nm}> ladder, Jacobian out (generic kernels, dense prime)", n2, timeit(lambda: e.scalar_mult(cv, k, P2[0], P2[1], flags=1, out=outj), 5), 555968, 192, "scalar mults")
    row(f"scalar_mult<{nm}> ladder, affine out", n2, timeit(lambda: e.scalar_mult(cv, k, P2[0], P2[1], flags=3, out=outj), 5), 555968 + 19 * 136, 160, "scalar mults")
    row(f"to_affine<{nm}> (simultaneous inversion)", n2, timeit(lambda: e.to_affine(cv, outj)), int((7 + 88 / 32) * 136), 256, "points")
    vw = int((66 + 4 + 63 * 44 + 4 + 7 + 88 / 32) * 136)              # k_gvarwin.hip: the table over one Z, a' = a Zg^4, 63 windows x (25M + 19S), Z' Zg + the leaving products, the shared inversion
    row(f"scalar_mult<{nm}> windowed variable base (the lane's table over one Z, modified Jacobian doublings), affine out", n2, timeit(lambda: e.scalar_mult(cv, k, b2x, b2y, flags=2 | 4, out=outj), 5), vw, 160, "scalar mults")
    row(f"scalar_mult<{nm}> windowed variable base, constant time (every entry read in every window), affine out", n2, timeit(lambda: e.scalar_mult(cv, k, b2x, b2y, flags=2 | 4 | 128, out=outj), 5), vw, 160, "scalar mults")
    u1 = e.fill_random(n2, SEED, 21, clear_top_bits=1)
    comb = 63 * 11 * 136                                               # the generator's 4-bit comb: 63 mixed additions (8M + 3S)
    row(f"scalar_mult_base<{nm}> windowed (the generator's comb in LDS), affine out", n2, timeit(lambda: e.scalar_mult_base(cv, k, flags=2 | 4, out=outj)), comb + int((7 + 88 / 32) * 136), 96, "scalar mults")
    comb5 = 51 * 11 * 136                                              # the constant-time 5-bit comb: 51 mixed additions
    row(f"scalar_mult_base<{nm}> windowed, constant time (5-bit windows, every entry of a window read), affine out", n2, timeit(lambda: e.scalar_mult_base(cv, k, flags=2 | 4 | 128, out=outj)), comb5 + int((7 + 88 / 32) * 136), 96, "scalar mults")
    comb7 = 36 * 11 * 136                                              # the signed 7-bit comb: 36 mixed additions
    row(f"scalar_mult_base<{nm}> signed 7-bit windows (the generator's comb, 148 KiB of LDS), affine out", n2, timeit(lambda: e.scalar_mult_base(cv, k, flags=2 | 8, out=outj)), comb7 + int((7 + 88 / 32) * 136), 96, "scalar mults")
    comb20 = 12 * 11 * 136                                             # the 20-bit comb in device memory: 12 mixed additions
    row(f"scalar_mult_base<{nm}> 20-bit windows (the generator's comb in device memory), affine out", n2, timeit(lambda: e.scalar_mult_base(cv, k, flags=2 | 32, out=outj)), comb20 + int((7 + 88 / 32) * 136), 96 + 832, "scalar mults")
    two = comb20 + vw + int(((7 + 88 / 32) + 6 + 88 / 32) * 136)        # the comb and its conversion, the window loop (its conversion inside vw), the affine addition with its shared inversion
    row(f"double_scalar_mult<{nm}> u1*G + u2*Q (the 20-bit comb + the lane's window table)", n2, timeit(lambda: e.double_scalar_mult(cv, u1, k, b2x, b2y), 5), two, 160, "verifications")
    rr = e.fill_random(n2, SEED, 22, clear_top_bits=1); ss = e.fill_random(n2, SEED, 23, clear_top_bits=1)
    row(f"ecdsa_verify<{nm}> (e, r, s, Q -> ok)", n2, timeit(lambda: e.ecdsa_verify(cv, u1, rr, ss, b2x, b2y), 5), two + 6 * 136, 161, "verifications")
    row(f"ecdsa_sign<{nm}> (e, d, k -> r, s: k G on the constant-time comb)", n2, timeit(lambda: e.ecdsa_sign(cv, u1, rr, ss), 5), 51 * 11 * 136 + int((4 + 88 / 32 + 9 + 7) * 136), 160, "signatures")
    wire = e.sec1_encode(cv, b2x, b2y, True)
    row(f"sec1_decode<{nm}> compressed (decompression: x^((p+1)/4) in sliding windows on 29-bit limbs)", n2, timeit(lambda: e.sec1_decode(cv, wire, True)), (253 + 64 + 4) * 136, 33 + 64, "points",
        executed_mad32=254 * 126 + 68 * 162 + 6 * 200)                   # sqr29 126 / mul29 162 multiply-adds with the dense reduction; right-hand side and check on canonical words
    wire = e.sec1_encode(cv, b2x, b2y, False)
    row(f"sec1_decode<{nm}> uncompressed (validation)", n2, timeit(lambda: e.sec1_decode(cv, wire, False)), 4 * 136, 65 + 64, "points")
    del k, s2, b2x, b2y, P2, outj, wire, u1, rr, ss
print(json.dumps(out, indent=1))
