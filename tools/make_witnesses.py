#!/usr/bin/env python3
"""tools/make_witnesses.py -- writes tests/golden/fe29_witnesses.json: concrete inputs at which the reduced-radix arithmetic of
ecsimd_amd/csrc/fe29.cuh meets the LARGEST columns and limbs its interval proofs allow, with the exact model's outputs.

  * "product" entries: for every interval proof (ladder, comb, window loop, GLV loop, complete law; P-256, secp256k1, and the any-prime ladder on three
    real dense primes) the products and squares whose columns come closest to 2^63, each with an operand pair whose limbs sit at the ends of the boxes the
    proof hands THAT call (tools/radix29_model.py product_witnesses: achieved / proven >= 0.99 on the built-in primes);
  * "function" entries: whole functions (zdau29, madd29, jdbl29, dbl_add29, gjdbl29, zaddu29, madd29v, pdbl29, padd29) on states at vertices of their loop invariants,
    found by hill-climbing on the exact model's worst column, plus random states inside the invariant box.

The file is DATA minted by this repo's own exact model (not by the reference -- the reference has no such representation); the exact model itself is held
to the big-int formulas by tests/test_radix29_model.py.  tests/test_gpu_witness.py runs every entry through ecsimd_hip_fe29_raw on the device and compares
limb for limb.    python tools/make_witnesses.py [--check]
"""
import json
import math
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import radix29_model as m  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "fe29_witnesses.json")
DENSE = {"brainpoolP256r1": m.BRAINPOOL_P256, "sm2": m.SM2_P, "frp256v1": m.FRP256_P}
CURVES = {"p256": m.CURVE_P256, "secp256k1": m.CURVE_SECP}
CURVES.update({k: m.Curve.dense(k, p) for k, p in DENSE.items()})
PROOFS = [("ladder", m.prove_invariant, ("p256", "secp256k1")), ("comb", m.prove_comb_invariant, ("p256", "secp256k1")), ("window", m.prove_window_invariant, ("p256", "secp256k1")),
          ("glv", m.prove_glv_invariant, ("secp256k1",)), ("complete", m.prove_complete_invariant, ("secp256k1",))]
FUNCTIONS = [("zdau", ("p256", "secp256k1", "brainpoolP256r1", "sm2", "frp256v1")), ("madd", ("p256", "secp256k1", "brainpoolP256r1", "sm2", "frp256v1")), ("jdbl", ("p256", "secp256k1")),
             ("dbl_add", ("p256", "secp256k1", "brainpoolP256r1", "sm2", "frp256v1")), ("gjdbl", ("brainpoolP256r1", "sm2", "frp256v1")), ("zaddu", ("secp256k1", "brainpoolP256r1", "sm2", "frp256v1")),
             ("maddv", ("p256", "secp256k1")), ("pdbl", ("secp256k1",)), ("padd", ("secp256k1",))]


def build():
    entries = []
    for proof, fn, kinds in PROOFS:
        for kind in kinds:
            for what, a, b, r, ach, proven in m.product_witnesses(CURVES[kind], m.proof_calls(fn), top=6, seed=7):
                entries.append({"kind": "product", "proof": proof, "curve": kind, "op": "mul" if what == "mul" else "sqr", "swap": 0,
                                "in": [a] + ([b] if b is not None else []), "out": [r], "worst_column": ach, "proven_column": proven})
    # the any-prime ladder proof: its boxes (every limb of p an interval) aimed at with three real dense primes
    any_calls = m.proof_calls(m.prove_invariant)(m.CURVE_ANY)
    for kind in DENSE:
        for what, a, b, r, ach, proven in m.product_witnesses(CURVES[kind], lambda cv: any_calls, top=4, seed=9):
            entries.append({"kind": "product", "proof": "ladder, any odd p < 2^256", "curve": kind, "op": "mul" if what == "mul" else "sqr", "swap": 0,
                            "in": [a] + ([b] if b is not None else []), "out": [r], "worst_column": ach, "proven_column": proven})
    # ... and the comb's (k_gcomb.hip: madd29 with the dense reduction), proven for any odd p as it stands
    any_comb = m.proof_calls(m.prove_comb_invariant)(m.CURVE_ANY)
    for kind in DENSE:
        for what, a, b, r, ach, proven in m.product_witnesses(CURVES[kind], lambda cv: any_comb, top=3, seed=11):
            entries.append({"kind": "product", "proof": "comb, any odd p < 2^256", "curve": kind, "op": "mul" if what == "mul" else "sqr", "swap": 0,
                            "in": [a] + ([b] if b is not None else []), "out": [r], "worst_column": ach, "proven_column": proven})
    # ... and the window loop and table of a registered curve (k_gvarwin.hip), proven for any odd p as well
    for proof, fn, seed in (("window loop, any odd p < 2^256", m.prove_gwindow_invariant, 13), ("window table, any odd p < 2^256", m.prove_gtable, 15)):
        calls = m.proof_calls(fn)(m.CURVE_ANY)
        for kind in DENSE:
            for what, a, b, r, ach, proven in m.product_witnesses(CURVES[kind], lambda cv: calls, top=3, seed=seed):
                entries.append({"kind": "product", "proof": proof, "curve": kind, "op": "mul" if what == "mul" else "sqr", "swap": 0,
                                "in": [a] + ([b] if b is not None else []), "out": [r], "worst_column": ach, "proven_column": proven})
    for op, kinds in FUNCTIONS:
        for kind in kinds:
            cv = CURVES[kind]
            for seed in range(3):
                st, out, worst = m.witness_search(op, cv, seed, steps=500, swap=bool(seed & 1))
                entries.append({"kind": "function", "curve": kind, "op": op, "swap": seed & 1, "in": st, "out": out, "worst_column": worst})
            # breadth: random states anywhere inside the invariant box (limbs uniform in their intervals, the value inside its own)
            rng = random.Random(1000 + len(entries))
            code, names, inv_fn, f, nout = m.OPS[op]
            inv = inv_fn(cv)
            got = 0
            while got < 8:
                st = []
                for n_ in names:
                    iv = inv[n_]
                    low = [rng.randint(*iv.l[i]) for i in range(m.NL - 1)]
                    S = sum(v << (m.W * i) for i, v in enumerate(low)); sh = m.W * (m.NL - 1)
                    lo = max(iv.l[m.NL - 1][0], -((-(iv.v[0] - S)) // (1 << sh))); hi = min(iv.l[m.NL - 1][1], (iv.v[1] - S) >> sh)
                    if lo > hi:
                        break
                    st.append(low + [rng.randint(lo, hi)])
                if len(st) != len(names):
                    continue
                E = m.Exact(cv)
                sw = bool(got & 1)
                out = f(E, [list(x) for x in st], sw)
                entries.append({"kind": "function", "curve": kind, "op": op, "swap": int(sw), "in": st, "out": out, "worst_column": E.worst_col})
                got += 1
    return {"_about": "generated by tools/make_witnesses.py from tools/radix29_model.py (this repo's exact model of fe29.cuh): operands at the extremes the interval proofs allow; "
                      "limbs are signed 32-bit integers, value = sum l[i] * 2^(29 i)", "curves": {k: format(p, "064x") for k, p in DENSE.items()}, "entries": entries}


if __name__ == "__main__":
    data = build()
    text = json.dumps(data, separators=(",", ":"))
    if "--check" in sys.argv:
        same = os.path.exists(OUT) and open(OUT).read() == text
        print("fe29_witnesses.json is", "up to date" if same else "STALE")
        sys.exit(0 if same else 1)
    open(OUT, "w").write(text)
    prod = [e for e in data["entries"] if e["kind"] == "product"]
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(data["entries"]), "entries;",
          "product witnesses reach %.4f .. %.4f of their call's proven column; the largest column met: 2^%.3f" % (
              min(e["worst_column"] / e["proven_column"] for e in prod), max(e["worst_column"] / e["proven_column"] for e in prod),
              math.log2(max(e["worst_column"] for e in data["entries"]))))
