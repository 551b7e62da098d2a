"""CPU suite, part 2: the C-ABI library builds, loads and exports every symbol include/ecsimd_hip.h
declares; the product path fails loudly (no CPU fallback) where there is no GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import ecsimd_amd
    if not os.path.exists(ecsimd_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    return ecsimd_amd.load_library()


def test_every_declared_symbol_is_exported(lib):
    from ecsimd_amd.engine import declared_symbols
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    # and nothing is exported that the header does not declare
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "ecsimd_amd", "libecsimd_hip.so")], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ecsimd_hip_\w+)", out))
    assert exported == set(syms), exported ^ set(syms)


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "ecsimd_hip.h"\nint main(void) { return sizeof(ecsimd_hip_ctx*) ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)


def test_constants_without_a_gpu(lib, oracle):
    """ecsimd_hip_get_constant is host-only: the engine's constants equal the oracle's / the reference's."""
    names = ["p", "a", "b", "gx", "gy", "r_p", "rsq_p", "pm1_r_p", "am", "bm", "p_m2", "p_sqrt"]
    for cv in (0, 1):
        c = oracle.constants(cv)
        for i, nm in enumerate(names):
            out = (C.c_uint64 * 4)()
            assert lib.ecsimd_hip_get_constant(C.c_int(cv), C.c_int(i), out) == 0
            assert list(out) == [int(v) for v in c[nm]], (cv, nm)
    assert lib.ecsimd_hip_get_constant(C.c_int(-1), C.c_int(0), (C.c_uint64 * 4)()) == -1
    assert lib.ecsimd_hip_get_constant(C.c_int(1 << 20), C.c_int(0), (C.c_uint64 * 4)()) == -1      # no such field id
    assert lib.ecsimd_hip_get_constant(C.c_int(0), C.c_int(12), (C.c_uint64 * 4)()) == -1


def test_modulus_registry_without_a_gpu(lib, oracle):
    """ecsimd_hip_register_modulus is host-only: what it derives from a modulus (R, R^2, -R mod p, p - 2, (p + 1)/4 -- mgry_csts.h:15-35)
    equals the oracle's for the same modulus; the same p gives the same id, the curve primes their curve ids, even values are refused."""
    import numpy as np
    from ecsimd_amd.engine import register_modulus, EcsimdHipError, P256_ORDER, SECP256K1_ORDER
    from oracle.loader import REF_MODULI, to_int
    rng = np.random.default_rng(8)
    moduli = list(REF_MODULI.values()) + [3, 5, 2**64 + 13, 2**256 - 189] + [int.from_bytes(rng.bytes(32), "big") | 1 for _ in range(8)]
    for p in moduli:
        fid = register_modulus(p)
        assert fid >= 2 and register_modulus(p) == fid
        # (p, PRIME) is another entry (the flag changes what gfp_inverse computes, so it must not change under a holder of the unflagged id) --
        # except for the built-in group orders, which ARE prime whatever the caller says
        builtin = fid in (P256_ORDER, SECP256K1_ORDER)
        fidp = register_modulus(p, prime=True)
        assert (fidp == fid) == builtin and register_modulus(p, prime=True) == fidp and register_modulus(p) == fid
        c = oracle.constants(oracle.register_modulus(p))
        for which, key in ((0, "p"), (5, "r_p"), (6, "rsq_p"), (7, "pm1_r_p"), (10, "p_m2"), (11, "p_sqrt")):
            out = (C.c_uint64 * 4)()
            assert lib.ecsimd_hip_get_constant(C.c_int(fid), C.c_int(which), out) == 0
            assert list(out) == [int(v) for v in c[key]], (hex(p), key)
        for which in (1, 2, 3, 4, 8, 9):                                                   # no curve behind a field id
            out = (C.c_uint64 * 4)(1, 1, 1, 1)
            assert lib.ecsimd_hip_get_constant(C.c_int(fid), C.c_int(which), out) == 0 and list(out) == [0, 0, 0, 0]
    assert register_modulus(REF_MODULI["n_p256"]) == P256_ORDER and register_modulus(REF_MODULI["n_secp256k1"]) == SECP256K1_ORDER
    assert register_modulus(0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff) == 0 and register_modulus(2**256 - 2**32 - 977) == 1
    for bad in (0, 1, 2, 2**255, 2**256 - 2):
        with pytest.raises(EcsimdHipError):
            register_modulus(bad)
    fid = C.c_int()
    assert lib.ecsimd_hip_register_modulus(None, C.c_int(0), C.byref(fid)) == -1
    assert lib.ecsimd_hip_register_modulus((C.c_uint64 * 4)(7, 0, 0, 0), C.c_int(2), C.byref(fid)) == -1      # unknown flag


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to run rather than compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ecsimd_amd
    with pytest.raises(ecsimd_amd.EcsimdHipError):
        ecsimd_amd.Engine(0)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    bad = []
    for base in ("ecsimd_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".cuh", ".hip", ".inc", ".cpp", "Makefile")):
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle[/.]|libecsimd_oracle|libecsimd_ref|/root/reference", text):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    nm = subprocess.run(["ldd", os.path.join(ROOT, "ecsimd_amd", "libecsimd_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in nm and "ecsimd_ref" not in nm


def test_group_shard_arithmetic_without_a_gpu():
    """ecsimd_hip_shard_range (the C ABI's partition of a batch over a device group) agrees with the Python runner's
    and covers [0, n) exactly; BASELINE configs[3]: 2^24 over 8 GPUs = 2^21 each.  No GPU call is made."""
    import ecsimd_amd
    from ecsimd_amd.shard import shard_range
    assert ecsimd_amd.shard_range_c(1 << 24, 5, 8) == (5 << 21, 1 << 21)
    for n in (0, 1, 7, 8, 9, 1000003, 1 << 22):
        for g in (1, 2, 3, 5, 8):
            spans = [ecsimd_amd.shard_range_c(n, m, g) for m in range(g)]
            assert spans == [shard_range(n, m, g) for m in range(g)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
    lib = ecsimd_amd.load_library()
    f, c = C.c_size_t(), C.c_size_t()
    assert lib.ecsimd_hip_shard_range(C.c_size_t(8), C.c_int(2), C.c_int(2), C.byref(f), C.byref(c)) == -1      # member out of range
    assert lib.ecsimd_hip_shard_range(C.c_size_t(8), C.c_int(0), C.c_int(0), C.byref(f), C.byref(c)) == -1
    g = C.c_void_p()
    assert lib.ecsimd_hip_group_init(None, C.c_int(1), C.byref(g)) == -1 and not g
    import torch
    if not torch.cuda.is_available():
        assert lib.ecsimd_hip_group_init((C.c_int * 1)(0), C.c_int(1), C.byref(g)) == -2 and not g               # no device: no fallback


def test_curve_registry_without_a_gpu(lib, oracle):
    """ecsimd_hip_register_curve is host-only (round 5: curve_group<Curve> for any Curve, curve.h:12-15): what it derives -- the field's constants, Am, Bm
    (curve_group.h:31-32) -- equals the oracle's for the same curve; the same parameters give the same id, the built-in curves their own ids unless the
    generic kernels are asked for; and what the reference could not instantiate or what is not a curve is refused: p != 3 mod 4 (gfp.h:84), a generator off
    the curve, a singular curve, a coordinate >= p, an even or zero order."""
    from ecsimd_amd.engine import register_curve, EcsimdHipError, FIRST_REGISTERED_CURVE
    from oracle.loader import REF_CURVES, to_int
    from helpers import CURVE_PARAMS, P256, SECP256K1
    seen = set()
    for name, c in REF_CURVES.items():
        cid = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"])
        assert cid >= FIRST_REGISTERED_CURVE and cid not in seen and register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"]) == cid
        seen.add(cid)
        want = oracle.constants(oracle.register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"]))
        for which, key in enumerate(["p", "a", "b", "gx", "gy", "r_p", "rsq_p", "pm1_r_p", "am", "bm", "p_m2", "p_sqrt"]):
            out = (C.c_uint64 * 4)()
            assert lib.ecsimd_hip_get_constant(C.c_int(cid), C.c_int(which), out) == 0
            assert list(out) == [int(v) for v in want[key]], (name, key)
    from test_gpu_curves import _random_curve                                  # random primes p = 3 mod 4 of every size: what the host derives for them
    for bits, kind in ((17, "random"), (31, "-3"), (64, "random"), (129, "0"), (255, "random"), (256, "random")):
        c = _random_curve(bits, 1000 * bits + len(kind), kind)
        cid = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"])
        want = oracle.constants(oracle.register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"]))
        for which, key in enumerate(["p", "a", "b", "gx", "gy", "r_p", "rsq_p", "pm1_r_p", "am", "bm", "p_m2", "p_sqrt"]):
            out = (C.c_uint64 * 4)()
            assert lib.ecsimd_hip_get_constant(C.c_int(cid), C.c_int(which), out) == 0
            assert list(out) == [int(v) for v in want[key]], (bits, key)
    for cv in (P256, SECP256K1):
        c = CURVE_PARAMS[cv]
        assert register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"]) == cv
        g = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"], generic_kernels=True)
        assert g >= FIRST_REGISTERED_CURVE and g not in seen and register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"], generic_kernels=True) == g
    c = REF_CURVES["brainpoolP256r1"]
    bad = [
        dict(c, gy=c["gy"] ^ 1),                                             # the generator is not on the curve
        dict(c, b=(c["b"] + 1) % c["p"]),                                    # ... nor on this one
        dict(c, gx=c["gx"] + c["p"]) if c["gx"] + c["p"] < 2**256 else dict(c, a=c["p"]),   # a coordinate >= p
        dict(p=2**255 - 19, a=486662, b=1, gx=9, gy=1),                      # p = 1 mod 4: the reference's GFp does not instantiate (gfp.h:84)
        dict(c, p=c["p"] + 1),                                               # even
        dict(p=c["p"], a=0, b=0, gx=0, gy=0),                                # singular (and G = (0, 0) is on it)
        dict(p=c["p"], a=c["p"] - 3, b=2, gx=1, gy=0),                       # y^2 = x^3 - 3x + 2 = (x - 1)^2 (x + 2): singular, G = (1, 0) on it
        dict(c, n=c["n"] + 1), dict(c, n=0),                                 # an even / zero order
    ]
    for b in bad:
        with pytest.raises(EcsimdHipError):
            register_curve(b["p"], b["a"], b["b"], b["gx"], b["gy"], b.get("n"))
    # the order is part of the key: what an id does depends on it (the comb, ECDSA), so a registration with another order -- or without one -- is another id and
    # never changes what an id somebody else holds does
    with_n = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"])
    without = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"])
    other = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"] + 2)
    assert len({with_n, without, other}) == 3 and register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"]) == with_n and register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"]) == without
    # what an id can do beyond the reference's layers is decided at registration, on the host (ecsimd_hip_curve_capabilities): the variable-base window loop
    # needs a group of PRIME order -- n in p's Hasse interval and a Miller-Rabin prime -- since only then does every valid point have order n
    from ecsimd_amd.engine import curve_capabilities, CURVE_HAS_ORDER, CURVE_COMB, CURVE_ECDSA, CURVE_WINDOW_VARIABLE_BASE
    everything = CURVE_HAS_ORDER | CURVE_COMB | CURVE_ECDSA | CURVE_WINDOW_VARIABLE_BASE
    assert curve_capabilities(P256) == everything and curve_capabilities(SECP256K1) == everything
    for name, rc in REF_CURVES.items():
        assert curve_capabilities(register_curve(rc["p"], rc["a"], rc["b"], rc["gx"], rc["gy"], rc["n"])) == everything, name
    assert curve_capabilities(with_n) == everything and curve_capabilities(without) == 0
    assert curve_capabilities(other) & (CURVE_HAS_ORDER | CURVE_COMB | CURVE_WINDOW_VARIABLE_BASE) == CURVE_HAS_ORDER | CURVE_COMB      # n + 2 = 3 * 5 * ...: in the interval, not a prime
    assert (c["n"] + 2) % 3 == 0
    far = register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], CURVE_PARAMS[P256]["n"])                                        # a prime, but 2^253 away from p + 1: not this group's order
    assert curve_capabilities(far) & (CURVE_COMB | CURVE_WINDOW_VARIABLE_BASE) == CURVE_COMB
    edge = 2 * int(c["p"] ** 0.5)                                                                                                  # the Hasse bound itself (float sqrt: within 2^-52 of it)
    for cv in (P256, SECP256K1):                                                                                                   # ... and the generic registration of the built-in curves
        bc = CURVE_PARAMS[cv]
        assert curve_capabilities(register_curve(bc["p"], bc["a"], bc["b"], bc["gx"], bc["gy"], bc["n"], generic_kernels=True)) == everything
    assert abs(c["n"] - c["p"] - 1) < edge
    with pytest.raises(EcsimdHipError):
        curve_capabilities(FIRST_REGISTERED_CURVE + 4000)
    cid = C.c_int()
    assert lib.ecsimd_hip_register_curve(None, None, None, None, None, None, C.c_int(0), C.byref(cid)) == -1
    assert lib.ecsimd_hip_get_constant(C.c_int(FIRST_REGISTERED_CURVE + 4000), C.c_int(0), (C.c_uint64 * 4)()) == -1
