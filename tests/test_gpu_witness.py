"""GPU: the inputs at which the interval proofs of tools/radix29_model.py reach their extremes, on the device.

tests/golden/fe29_witnesses.json (tools/make_witnesses.py) holds operand pairs whose 64-bit columns sit AT the bound their proof derives for that very
product (up to 2^63.000 -- the last representable magnitude), whole loop states at vertices of the invariants, and random states inside them, each with
the exact model's output limbs.  ecsimd_hip_fe29_raw runs the device function on those raw limbs: every output limb must equal the model's -- an overflowing
column, a carry pass missing from the device or a wrong reduction constant shows up as a different limb.  (The model is tied to the device source
structurally and to the big-int formulas numerically by tests/test_radix29_model.py; no other entry point can feed the loops anything but tight limbs.)"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPCODE = {"zdau": 0, "madd": 1, "jdbl": 2, "dbl_add": 3, "maddv": 4, "pdbl": 5, "padd": 6, "mul": 7, "sqr": 8, "gjdbl": 9, "zaddu": 10}


def test_the_device_on_the_proofs_extreme_inputs(engine):
    import torch
    from ecsimd_amd.engine import register_curve
    from oracle.loader import REF_CURVES
    data = json.load(open(os.path.join(ROOT, "tests", "golden", "fe29_witnesses.json")))
    cid = {"p256": 0, "secp256k1": 1}
    for name, c in REF_CURVES.items():
        cid[name] = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"])
        assert format(c["p"], "064x") == data["curves"][name]
    groups = {}
    for e in data["entries"]:
        groups.setdefault((e["curve"], e["op"], e["swap"]), []).append(e)
    checked, at_the_bound = 0, 0
    for (curve, op, swap), es in sorted(groups.items()):
        inp = torch.tensor(np.array([e["in"] for e in es], dtype=np.int64).astype(np.int32), device=engine.tdev).contiguous()
        got = engine.fe29_raw(cid[curve], OPCODE[op], inp, swap).cpu().numpy()
        exp = np.array([e["out"] for e in es], dtype=np.int64).astype(np.int32)
        bad = np.flatnonzero((got != exp).reshape(len(es), -1).any(axis=1))
        assert bad.size == 0, (curve, op, swap, bad[:4], [es[i].get("proof") for i in bad[:4]])
        checked += len(es)
        at_the_bound += sum(1 for e in es if e["kind"] == "product" and e["worst_column"] >= 0.9375 * e["proven_column"] and e["worst_column"] >= 2**62)
    assert checked == len(data["entries"]) >= 200 and at_the_bound >= 20
    # a function that does not exist for a curve is refused, not served by another curve's code
    from ecsimd_amd import EcsimdHipError
    with pytest.raises(EcsimdHipError):
        engine.fe29_raw(0, OPCODE["pdbl"], torch.zeros((1, 3, 9), dtype=torch.int32, device=engine.tdev))
    with pytest.raises(EcsimdHipError):
        engine.fe29_raw(cid["sm2"], OPCODE["jdbl"], torch.zeros((1, 3, 9), dtype=torch.int32, device=engine.tdev))     # (a registered curve doubles with gjdbl29)
    with pytest.raises(EcsimdHipError):
        engine.fe29_raw(0, OPCODE["gjdbl"], torch.zeros((1, 4, 9), dtype=torch.int32, device=engine.tdev))
