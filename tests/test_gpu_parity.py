"""GPU suite (-m gpu): the HIP path, called through the C ABI, against
  * the reference-minted fixtures and the reference's own KATs (tests/golden/),
  * the CPU oracle on seeded lane-distinct inputs (sizes the oracle finishes in seconds),
  * size-independent properties at BASELINE.json's full batch sizes.
Bit-exact everywhere (integer arithmetic): no tolerances."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import (CURVE_PARAMS, CURVE_NAMES, P256, SECP256K1, SEED, hexes_to_arr, arr_to_hexes, from_int, to_int, ints_to_arr,
                     arr_to_ints, from_hex, fill_random_np, ec_mul, ec_add, jacobian_mgry_to_affine_int)
from test_oracle import run_against_golden, structured_words
from ecsimd_amd import ALG_NO_ENDOMORPHISM, OUT_AFFINE, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG, ALG_CONSTANT_TIME, REF_SQUARE_COMPAT, GROUP_NO_GATHER, EcsimdHipError

pytestmark = pytest.mark.gpu
CURVES = [P256, SECP256K1]
THREADS = min(16, os.cpu_count() or 1)


@pytest.fixture(scope="module")
def gpu(engine):
    from gpu_adapter import EngineNP
    return EngineNP(engine)


def test_native_library_is_loaded(engine):
    import ecsimd_amd
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(ecsimd_amd.lib_path()) in maps, "libecsimd_hip.so is not mapped: the HIP path is not the one running"
    assert b"gfx950" in engine.lib.ecsimd_hip_version()


def test_fixtures_from_the_reference(gpu, golden):
    run_against_golden(gpu, golden)


def test_reference_kats_on_gpu(gpu, kats):
    k = kats["p256"]; cv = P256; c = CURVE_PARAMS[cv]
    gx, gy = ints_to_arr([c["gx"]] * 4), ints_to_arr([c["gy"]] * 4)          # n = 4 = one eve::wide of the reference
    G = gpu.from_affine(cv, gx, gy)
    aff = lambda J: tuple(format(to_int(v[0]), "064x") for v in gpu.to_affine(cv, J))
    dbl, Gu = gpu.dblu(cv, G)
    assert np.array_equal(dbl[2], Gu[2]) and aff(dbl) == (k["2G"]["x"], k["2G"]["y"]) and aff(Gu) == aff(G)
    tri, Gu2 = gpu.zaddu(cv, Gu, dbl)
    assert aff(tri) == (k["3G"]["x"], k["3G"]["y"])
    five, _ = gpu.zdau(cv, dbl, Gu)
    assert aff(five) == (k["5G"]["x"], k["5G"]["y"])
    for s in k["scalar_mult"]:
        kk = np.tile(from_hex(s["k"]), (4, 1))
        for J in (gpu.scalar_mult(cv, kk, gx, gy), gpu.scalar_mult_1s(cv, from_hex(s["k"]), gx, gy), gpu.scalar_mult_base(cv, kk)):
            assert aff(J) == (s["x"], s["y"]), s["src"]
        ax, ay = gpu.scalar_mult(cv, kk, gx, gy, affine=True)[:2]
        assert format(to_int(ax[3]), "064x") == s["x"] and format(to_int(ay[3]), "064x") == s["y"]
    y, ok = gpu.compute_y(cv, hexes_to_arr([k["from_x"]["x"]] * 4))
    assert ok.tolist() == [1] * 4 and format(to_int(y[0]), "064x") == k["from_x"]["y"]
    km = kats["mgry_secp256k1"]; cv = SECP256K1
    vals = hexes_to_arr(km["from_to_roundtrip"]["values"])
    assert np.array_equal(gpu.mgry_to_classical(cv, gpu.mgry_from_classical(cv, vals)), vals)
    ma = gpu.mgry_from_classical(cv, hexes_to_arr([km["ops"]["a"]]))
    for c_ in km["ops"]["pow"]:
        assert format(to_int(gpu.mgry_to_classical(cv, gpu.mgry_pow(cv, ma, from_hex(c_["e"])))[0]), "064x") == c_["r"]
    s, ok = gpu.gfp_sqrt(cv, gpu.mgry_from_classical(cv, hexes_to_arr([km["gfp"]["sqrt"]["a"]])))
    assert ok[0] == 1 and format(to_int(gpu.mgry_to_classical(cv, s)[0]), "064x") == km["gfp"]["sqrt"]["r"]


def test_constants_match_the_oracle(engine, oracle):
    names = ["p", "a", "b", "gx", "gy", "r_p", "rsq_p", "pm1_r_p", "am", "bm", "p_m2", "p_sqrt"]
    for cv in CURVES:
        c = oracle.constants(cv)
        for i, nm in enumerate(names):
            assert np.array_equal(engine.constant(cv, i), c[nm]), (cv, nm)


@pytest.mark.parametrize("n", [0, 1, 3, 63, 64, 65, 255, 256, 257, 1000])
def test_ragged_and_empty_batches(gpu, oracle, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); b = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] >>= np.uint64(1); b[:, 3] >>= np.uint64(1)       # < 2^255 < p for both primes
    for cv in CURVES:
        assert np.array_equal(gpu.mgry_mul(cv, a, b), oracle.mgry_mul(cv, a, b))
        assert np.array_equal(gpu.mod_sub(cv, a, b), oracle.mod_sub(cv, a, b))
    assert np.array_equal(gpu.mul(a, b), oracle.mul(a, b))


@pytest.mark.parametrize("cv", CURVES)
def test_field_ops_vs_oracle(gpu, oracle, cv):
    n = 20000; p = CURVE_PARAMS[cv]["p"]; rng = np.random.default_rng(11 + cv)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); b = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] >>= np.uint64(1); b[:, 3] >>= np.uint64(1)
    # edge values: 0, 1, p-1, p-2, 2^255-1 and values whose words are all-ones / carry-heavy
    edge = [0, 1, 2, p - 1, p - 2, p - 3, (p - 1) // 2, (p + 1) // 2, 2**255 - 1, 2**128 - 1, 2**224, 2**96 - 1, 2**192 + 2**96]
    for i, v in enumerate(edge):
        a[i] = from_int(v % p); b[len(edge) - 1 - i] = from_int(v % p)
    a[40:80] = ints_to_arr([(p - 1 - j) for j in range(40)]); b[40:80] = ints_to_arr([(p - 1 - 3 * j) for j in range(40)])
    for name in ("mod_add", "mod_sub", "mgry_mul"):
        assert np.array_equal(getattr(gpu, name)(cv, a, b), getattr(oracle, name)(cv, a, b)), name
    for name in ("mgry_sqr", "mgry_from_classical", "mgry_to_classical", "gfp_opposite"):
        assert np.array_equal(getattr(gpu, name)(cv, a), getattr(oracle, name)(cv, a)), name
    for cnt in (1, 2, 3, 7):
        assert np.array_equal(gpu.mod_shift_left(cv, a, cnt), oracle.mod_shift_left(cv, a, cnt))
    # the fused quadrupling the point formulas use (ECSIMD_HIP_SHIFT_FUSED): same residue for canonical operands -- the edge
    # values above, every top-two-bit pattern, and the neighbourhoods of k*p/4 where the conditional subtraction flips
    q4 = [(kq * p) // 4 + d for kq in (1, 2, 3) for d in (-2, -1, 0, 1, 2)] + [2**254 - 1, 2**254, 2**255 - 1, 2**255, 3 * 2**254 - 1, 3 * 2**254, p - 1, p - 2, 0, 1]
    af = a.copy(); af[100:100 + len(q4)] = ints_to_arr([v % p for v in q4])
    for cnt in (2, 3, 4, 7):
        assert np.array_equal(gpu.mod_shift_left(cv, af, cnt | 0x100), oracle.mod_shift_left(cv, af, cnt)), cnt
    sw = _carry_heavy_field_elements(cv, 20000, 9)
    assert np.array_equal(gpu.mod_shift_left(cv, sw, 2 | 0x100), oracle.mod_shift_left(cv, sw, 2))
    t8 = rng.integers(0, 2**64, size=(n, 8), dtype=np.uint64)
    t8[:, 7] &= np.uint64(2**63 - 1)                                   # < 2^511 < p * 2^256
    t8[0] = 0; t8[1] = ints_to_arr([p * (2**256) - 1], 8)[0]; t8[2] = ints_to_arr([(p - 1) * (p - 1)], 8)[0]
    assert np.array_equal(gpu.mgry_reduce(cv, t8), oracle.mgry_reduce(cv, t8))
    # non-canonical input to from_classical (n >= p), as tests/ops.cpp:232 feeds mod_add
    big = ints_to_arr([p, p + 1, 2**256 - 1, 2**256 - 2])
    assert np.array_equal(gpu.mgry_from_classical(cv, big), oracle.mgry_from_classical(cv, big))
    m = 512
    assert np.array_equal(gpu.gfp_inverse(cv, a[:m]), oracle.gfp_inverse(cv, a[:m]))
    s, ok = gpu.gfp_sqrt(cv, a[:m]); so, oko = oracle.gfp_sqrt(cv, a[:m])
    assert np.array_equal(ok, oko) and np.array_equal(s, so) and 0 < ok.sum() < m     # residues and non-residues both present
    e = from_int(0x1234567890abcdef_0fedcba987654321_00000000ffffffff_8000000000000001)
    assert np.array_equal(gpu.mgry_pow(cv, a[:m], e), oracle.mgry_pow(cv, a[:m], e))
    for e in (from_int(0), from_int(1), from_int(2**255)):
        assert np.array_equal(gpu.mgry_pow(cv, a[:64], e), oracle.mgry_pow(cv, a[:64], e))


def test_bignum_ops_vs_oracle(gpu, oracle):
    n = 20000; rng = np.random.default_rng(5)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); b = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[0] = from_int(2**256 - 1); b[0] = from_int(2**256 - 1); a[1] = from_int(0); b[2] = from_int(0); a[3] = b[3]
    a[4] = from_int(2**256 - 1); b[4] = from_int(1)
    for name in ("add", "sub"):
        s, f = getattr(gpu, name)(a, b); so, fo = getattr(oracle, name)(a, b)
        assert np.array_equal(s, so) and np.array_equal(f, fo), name
    s, f = gpu.shift_left_one(a); so, fo = oracle.shift_left_one(a)
    assert np.array_equal(s, so) and np.array_equal(f, fo)
    assert np.array_equal(gpu.sub_if_above(a, b), oracle.sub_if_above(a, b))
    assert np.array_equal(gpu.mul(a, b), oracle.mul(a, b)) and np.array_equal(gpu.square(a), oracle.square(a))


def test_carry_heavy_operands(gpu, oracle, golden):
    """Digits of 0 / ff..f / 80..0 / 7f..f: every carry chain saturates.  This is the operand class on
    which the REFERENCE's square() drops a carry (mul.h:186-190, "TODO: carry?" :207 -- pinned in
    tests/golden ref_vectors.json "square_defect"); the HIP path must return the exact value a*a."""
    a = structured_words(200000); b = np.roll(a, 7, axis=0)
    sq = gpu.square(a)
    assert np.array_equal(sq, oracle.square(a)) and np.array_equal(sq, gpu.mul(a, a))
    assert np.array_equal(gpu.mul(a, b), oracle.mul(a, b))
    for name in ("add", "sub"):
        s, f = getattr(gpu, name)(a, b); so, fo = getattr(oracle, name)(a, b)
        assert np.array_equal(s, so) and np.array_equal(f, fo)
    d = golden["square_defect"]; x = hexes_to_arr(d["a"])
    assert arr_to_hexes(gpu.square(x), 8) == d["exact_square"]
    assert arr_to_hexes(gpu.mgry_sqr(P256, x)) == d["p256_exact_mgry_sqr"]
    for cv in CURVES:
        p = CURVE_PARAMS[cv]["p"]; m = 20000
        ar = ints_to_arr([v % p for v in arr_to_ints(a[:m])]); br = ints_to_arr([v % p for v in arr_to_ints(b[:m])])
        assert np.array_equal(gpu.mgry_sqr(cv, ar), oracle.mgry_sqr(cv, ar))
        assert np.array_equal(gpu.mgry_sqr(cv, ar), gpu.mgry_mul(cv, ar, ar))
        for nm in ("mgry_mul", "mod_add", "mod_sub"):
            assert np.array_equal(getattr(gpu, nm)(cv, ar, br), getattr(oracle, nm)(cv, ar, br)), nm
        assert np.array_equal(gpu.mod_shift_left(cv, ar, 3), oracle.mod_shift_left(cv, ar, 3))
        # classical a*b mod p: on secp256k1 this is the pseudo-Mersenne reduction the ladder runs on
        ai, bi = arr_to_ints(ar), arr_to_ints(br)
        assert arr_to_ints(gpu.mod_mul(cv, ar, br)) == [x * y % p for x, y in zip(ai, bi)]
        worst = ints_to_arr([p - 1, p - 1, p - 2, 2**255, 2**128, (p - 1) // 2, 2**32 + 977, p - 2**32, 1, 0])
        wi = arr_to_ints(worst)
        assert arr_to_ints(gpu.mod_mul(cv, worst, worst[::-1].copy())) == [x * y % p for x, y in zip(wi, wi[::-1])]
        assert arr_to_ints(gpu.mod_mul(cv, worst, worst)) == [x * x % p for x in wi]
        mm = 2000
        assert np.array_equal(gpu.gfp_inverse(cv, ar[:mm]), oracle.gfp_inverse(cv, ar[:mm]))
        sg, okg = gpu.gfp_sqrt(cv, ar[:mm]); so, oko = oracle.gfp_sqrt(cv, ar[:mm])
        assert np.array_equal(okg, oko) and np.array_equal(sg, so)
        t8 = np.concatenate([a[:m], b[:m]], axis=1); t8[:, 7] &= np.uint64(2**63 - 1)
        assert np.array_equal(gpu.mgry_reduce(cv, t8), oracle.mgry_reduce(cv, t8))
        # carry-heavy points are not on the curve, but the formulas are polynomial maps: still bit-exact
        P = (ar[:4096], br[:4096], np.tile(oracle.constants(cv)["r_p"], (4096, 1)))
        (R, Pu), (Rg, Pug) = oracle.trplu(cv, P), gpu.trplu(cv, P)
        assert all(np.array_equal(u, v) for u, v in zip(Rg + Pug, R + Pu))
        (Rz, Qu), (Rzg, Qug) = oracle.zdau(cv, R, Pu), gpu.zdau(cv, R, Pu)
        assert all(np.array_equal(u, v) for u, v in zip(Rzg + Qug, Rz + Qu))
        Ra, Rag = oracle.add_z2_1(cv, Rz, (ar[:4096], br[:4096])), gpu.add_z2_1(cv, Rz, (ar[:4096], br[:4096]))
        assert all(np.array_equal(u, v) for u, v in zip(Rag, Ra))


def test_non_canonical_operands_behave_like_the_reference(gpu, oracle):
    """Operands >= p (the reference never rejects them: tests/ops.cpp:232 feeds mod_add an unreduced value).
    Every op performs ONE conditional subtraction like the reference, so the results -- possibly
    non-canonical -- must still be identical."""
    a = structured_words(50000, seed=3); b = np.roll(structured_words(50000, seed=4), 11, axis=0)      # anything in [0, 2^256)
    a[:4] = ints_to_arr([2**256 - 1] * 4); b[:2] = ints_to_arr([2**256 - 1, 2**256 - 2])
    for cv in CURVES:
        for nm in ("mod_add", "mod_sub", "mgry_mul"):
            assert np.array_equal(getattr(gpu, nm)(cv, a, b), getattr(oracle, nm)(cv, a, b)), (cv, nm)
        for nm in ("mgry_sqr", "mgry_from_classical", "mgry_to_classical", "gfp_opposite"):
            assert np.array_equal(getattr(gpu, nm)(cv, a), getattr(oracle, nm)(cv, a)), (cv, nm)
        assert np.array_equal(gpu.mod_shift_left(cv, a, 2), oracle.mod_shift_left(cv, a, 2))
        t8 = np.concatenate([a, b], axis=1)                                                            # up to 2^512 - 1
        assert np.array_equal(gpu.mgry_reduce(cv, t8), oracle.mgry_reduce(cv, t8))


def test_config0_ops_bench_batch8_on_gpu(gpu, oracle):
    """BASELINE configs[0] shapes (benchs/ops.cpp:106-116, batch = 8) through the HIP path."""
    n = 8; rng = np.random.default_rng(8)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); b = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    t8 = rng.integers(0, 2**64, size=(n, 8), dtype=np.uint64)
    a0 = a.copy(); a0[:, 0] &= np.uint64(0xFFFFFFFFFFFFFF00); t8[:, 0] &= np.uint64(0xFFFFFFFFFFFFFF00)
    s, c = gpu.add(a, b); so, co = oracle.add(a, b)
    assert np.array_equal(s, so) and np.array_equal(c, co)
    assert np.array_equal(gpu.mul(a, b), oracle.mul(a, b)) and np.array_equal(gpu.square(a), oracle.square(a))
    assert np.array_equal(gpu.mgry_sqr(SECP256K1, a0), oracle.mgry_sqr(SECP256K1, a0))
    assert np.array_equal(gpu.mgry_reduce(SECP256K1, t8), oracle.mgry_reduce(SECP256K1, t8))


def test_cmp_and_swap(engine, oracle):
    n = 1000; rng = np.random.default_rng(9)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); b = a.copy(); b[::2] = rng.integers(0, 2**64, size=(n // 2, 4), dtype=np.uint64)
    da, db = engine.to_device(a), engine.to_device(b)
    lt = engine.to_numpy(engine.cmp_lt(da, db))
    assert lt.tolist() == [int(to_int(x) < to_int(y)) for x, y in zip(a, b)]          # cmp.h:11-13
    import torch
    mask = torch.from_numpy((rng.integers(0, 2, size=n) * 255).astype(np.uint8)).to(da.device)
    engine.swap_if(mask, da, db)                                                        # tests/ops.cpp:179-208
    m = mask.cpu().numpy().astype(bool)
    assert np.array_equal(engine.to_numpy(da), np.where(m[:, None], b, a)) and np.array_equal(engine.to_numpy(db), np.where(m[:, None], a, b))
    # ifelse.h:15-22 if_else(m, x, y) = m ? x : y, into a fresh array and in place over either operand
    xa, xb = engine.to_device(a), engine.to_device(b)
    assert np.array_equal(engine.to_numpy(engine.if_else(mask, xa, xb)), np.where(m[:, None], a, b))
    import ctypes as C
    engine._call("if_else", engine._ptr(mask, 0), engine._ptr(xa), engine._ptr(xb), engine._ptr(xa), C.c_size_t(n))       # out aliases a
    assert np.array_equal(engine.to_numpy(xa), np.where(m[:, None], a, b))
    xa = engine.to_device(a)
    engine._call("if_else", engine._ptr(mask, 0), engine._ptr(xa), engine._ptr(xb), engine._ptr(xb), C.c_size_t(n))       # out aliases b
    assert np.array_equal(engine.to_numpy(xb), np.where(m[:, None], a, b))


@pytest.mark.parametrize("cv", CURVES)
def test_batched_inversion(engine, oracle, cv):
    """gfp_inverse: Montgomery's simultaneous inversion when out != a, one addition chain per element in place; both equal
    the oracle's a^(p-2), zeros map to zero and do not disturb their neighbours; ragged sizes around the per-lane batching."""
    import torch
    import ctypes as C
    for n in (1, 63, 64, 65, (1 << 17) + 3, (1 << 20) + 77):
        a = fill_random_np(n, SEED, 85, clear_top_bits=1)
        for i in (0, n // 2, n - 1):
            a[i] = 0
        am = engine.mgry_from_classical(cv, engine.to_device(a))
        inv = engine.gfp_inverse(cv, am)
        inplace = am.clone()
        assert engine.lib.ecsimd_hip_gfp_inverse(engine.ctx, C.c_int(cv), C.c_void_p(inplace.data_ptr()), C.c_void_p(inplace.data_ptr()), C.c_size_t(n)) == 0
        assert torch.equal(inv, inplace)
        idx = np.unique(np.concatenate([np.arange(min(n, 128)), np.arange(max(0, n - 128), n), np.array([n // 2])]))
        exp = oracle.gfp_inverse(cv, engine.to_numpy(am)[idx])
        assert np.array_equal(engine.to_numpy(inv)[idx], exp)
        prod = engine.mgry_mul(cv, inv, am)                                       # a * a^-1 = 1 (Montgomery form) except where a = 0
        one = engine.to_numpy(engine.mgry_from_classical(cv, engine.to_device(ints_to_arr([1]))))[0]
        pn = engine.to_numpy(prod); zero = (a == 0).all(axis=1)
        assert (pn[~zero] == one).all() and (pn[zero] == 0).all()


def test_lane_masks_on_the_device(engine):
    """cmp_eq (4- and 8-limb elements), mask NOT / AND / OR / EQ and the count behind all() / any()."""
    import torch
    n = 100_003
    a = fill_random_np(n, SEED, 81); b = a.copy()
    diff = np.arange(0, n, 7); b[diff, 3] ^= np.uint64(1 << 63)                    # differ in the top bit of the top limb
    b[5, 0] ^= np.uint64(1)
    eq = engine.cmp_eq(engine.to_device(a), engine.to_device(b))
    exp = (a == b).all(axis=1)
    assert np.array_equal(engine.to_numpy(eq) != 0, exp)
    a8 = np.concatenate([a, a[::-1]], axis=1); b8 = np.concatenate([b, a[::-1]], axis=1)
    eq8 = engine.cmp_eq(engine.to_device(a8), engine.to_device(b8))
    assert torch.equal(eq8, eq)
    lt = engine.cmp_lt(engine.to_device(a), engine.to_device(b))
    ltn = engine.to_numpy(lt) != 0
    for op, fn in ((0, lambda x, y: ~x), (1, np.logical_and), (2, np.logical_or), (3, lambda x, y: x == y)):
        got = engine.mask_op(op, eq, None if op == 0 else lt)
        assert np.array_equal(engine.to_numpy(got) != 0, fn(exp, ltn)), op
    assert engine.mask_count(eq) == int(exp.sum())
    assert engine.mask_count(engine.flags(n)) == 0 and engine.mask_count(engine.mask_op(0, engine.flags(n))) == n
    assert engine.mask_count(engine.flags(0)) == 0


def _lane_distinct_points(gpu, cv, n, stream=2):
    """P_i = s_i * G, affine classical, from the synthetic generator."""
    s = fill_random_np(n, SEED, stream)
    return gpu.scalar_mult_base(cv, s, affine=True)[:2]


@pytest.mark.parametrize("cv", CURVES)
def test_point_formulas_vs_oracle(gpu, oracle, cv):
    n = 4096
    bx, by = _lane_distinct_points(gpu, cv, n)
    P = oracle.from_affine(cv, bx, by)
    assert all(np.array_equal(u, v) for u, v in zip(gpu.from_affine(cv, bx, by), P))
    (R, Pu), (Rg, Pug) = oracle.dblu(cv, P), gpu.dblu(cv, P)
    assert all(np.array_equal(u, v) for u, v in zip(Rg + Pug, R + Pu)), "DBLU"
    (R3, Pu2), (R3g, Pu2g) = oracle.zaddu(cv, Pu, R), gpu.zaddu(cv, Pu, R)
    assert all(np.array_equal(u, v) for u, v in zip(R3g + Pu2g, R3 + Pu2)), "ZADDU"
    (Rt, Put), (Rtg, Putg) = oracle.trplu(cv, P), gpu.trplu(cv, P)
    assert all(np.array_equal(u, v) for u, v in zip(Rtg + Putg, Rt + Put)), "TRPLU"
    assert all(np.array_equal(u, v) for u, v in zip(Rt, R3)), "TRPLU == DBLU;ZADDU"
    (Rz, Qu), (Rzg, Qug) = oracle.zdau(cv, Rt, Put), gpu.zdau(cv, Rt, Put)
    assert all(np.array_equal(u, v) for u, v in zip(Rzg + Qug, Rz + Qu)), "ZDAU"
    assert np.array_equal(Rzg[2], Qug[2]), "co-Z invariant"
    Ra, Rag = oracle.add_z2_1(cv, Rz, (P[0], P[1])), gpu.add_z2_1(cv, Rz, (P[0], P[1]))
    assert all(np.array_equal(u, v) for u, v in zip(Rag, Ra)), "ADD_Z2_1"
    m = 512
    Jm = tuple(v[:m] for v in Ra)
    assert all(np.array_equal(u, v) for u, v in zip(gpu.to_affine(cv, Jm), oracle.to_affine(cv, Jm))), "to_affine"
    # 7P + P = 8P: check a few lanes against the independent affine model
    ax, ay = gpu.to_affine(cv, tuple(v[:8] for v in Ra))
    for i in range(8):
        assert (to_int(ax[i]), to_int(ay[i])) == ec_mul(cv, 8, (to_int(bx[i]), to_int(by[i])))
    y, ok = gpu.compute_y(cv, bx[:m])
    assert ok.all()
    p = CURVE_PARAMS[cv]["p"]
    assert all(to_int(u) in (to_int(v), p - to_int(v)) for u, v in zip(y, by[:m]))
    yo, oko = oracle.compute_y(cv, bx[:m])
    assert np.array_equal(y, yo) and np.array_equal(ok, oko)


@pytest.mark.parametrize("cv", CURVES)
def test_reduced_radix_zdau_vs_oracle(gpu, oracle, cv):
    """The ladder's loop body in the representation it runs in since round 4 -- nine signed 29-bit limbs, lazy carries, Montgomery radix
    2^261 (fe29.cuh) -- against the oracle's ZDAU (curve_group.h:120-153) iterated on the CPU: ecsimd_hip_zdau_repeat with radix 29 and
    with radix 32 (round 3's canonical 8-word form) must both return the oracle's X, Y of both points and their Z bit for bit, after
    1, 2, 7 and 67 iterations, with and without the per-iteration exchange of the two outputs, on curve points (TRPLU's co-Z pairs) and on
    carry-heavy digit-pattern coordinates (level J does not need curve points).  tools/radix29_model.py proves the bounds; this pins the bits."""
    from test_oracle import digit_pattern_operands
    n = 1536
    bx, by = _lane_distinct_points(gpu, cv, n, stream=9)
    Rt, Put = oracle.trplu(cv, oracle.from_affine(cv, bx, by))
    a = digit_pattern_operands()[::1093]
    pp = np.tile(oracle.constants(cv)["p"], (len(a), 1))
    red = lambda v: oracle.sub_if_above(v, pp)
    cases = [(Rt, Put), ((red(a), red(np.roll(a, 5, axis=0)), red(np.roll(a, 11, axis=0))), (red(np.roll(a, 17, axis=0)), red(np.roll(a, 23, axis=0)), red(np.roll(a, 11, axis=0))))]
    for P0, Q0 in cases:
        for iters, swap in ((1, 0), (1, 1), (2, 0b10), (7, 0b1011001), (67, 0xdeadbeefcafef00d)):
            P, Q = P0, Q0
            for t in range(iters):
                R, Qn = oracle.zdau(cv, P, Q)
                P, Q = (Qn, R) if (swap >> (t & 63)) & 1 else (R, Qn)
            exp = (P[0], P[1], Q[0], Q[1], P[2])
            assert np.array_equal(P[2], Q[2])
            for radix in (29, 32):
                got = gpu.zdau_repeat(cv, P0, (Q0[0], Q0[1]), iters, swap, radix)
                bad = [name for name, g, e in zip(("x1", "y1", "x2", "y2", "z"), got, exp) if not np.array_equal(g, e)]
                assert not bad, (radix, iters, hex(swap), bad)


@pytest.mark.parametrize("cv", CURVES)
def test_point_formulas_on_digit_pattern_coordinates(gpu, oracle, cv):
    """DBLU / ZADDU / ZDAU / ADD_Z2_1 as expression DAGs over GF(p) (parity level J does not need curve points): coordinates from the
    carry-heavy digit-pattern family, every 7th operand (239 946 points), so that the formulas' add / subtract / double chains and
    the fused forms (one reduction for a difference of products, quadrupling in one pass) see saturated words everywhere."""
    from test_oracle import digit_pattern_operands
    a = digit_pattern_operands()[::7]
    p = np.tile(oracle.constants(cv)["p"], (len(a), 1))
    x, y = oracle.sub_if_above(a, p), oracle.sub_if_above(np.roll(a, 1234, axis=0), p)
    P = oracle.from_affine(cv, x, y)
    (R, Pu), (Rg, Pug) = oracle.dblu(cv, P), gpu.dblu(cv, P)
    assert all(np.array_equal(u, v) for u, v in zip(Rg + Pug, R + Pu)), "DBLU"
    (R3, Pu2), (R3g, Pu2g) = oracle.zaddu(cv, Pu, R), gpu.zaddu(cv, Pu, R)
    assert all(np.array_equal(u, v) for u, v in zip(R3g + Pu2g, R3 + Pu2)), "ZADDU"
    (Rz, Qu), (Rzg, Qug) = oracle.zdau(cv, R3, Pu2), gpu.zdau(cv, R3, Pu2)
    assert all(np.array_equal(u, v) for u, v in zip(Rzg + Qug, Rz + Qu)), "ZDAU"
    # ZDAU again on its own output, with the raw patterns as a co-Z partner: arbitrary field elements in every slot
    Q = (x, y, Rz[2])
    (Rz2, Qu2), (Rz2g, Qu2g) = oracle.zdau(cv, Rz, Q), gpu.zdau(cv, Rz, Q)
    assert all(np.array_equal(u, v) for u, v in zip(Rz2g + Qu2g, Rz2 + Qu2)), "ZDAU on arbitrary field elements"
    Ra, Rag = oracle.add_z2_1(cv, Rz2, (x, y)), gpu.add_z2_1(cv, Rz2, (x, y))
    assert all(np.array_equal(u, v) for u, v in zip(Rag, Ra)), "ADD_Z2_1"


@pytest.mark.parametrize("cv", CURVES)
def test_ladder_on_digit_pattern_scalars_and_coordinates(gpu, oracle, cv):
    """The whole ladder (curve_group.h:189-218) at level J on structured inputs: 16 796 scalars from the digit-pattern family (runs of
    ones and zeros, lone top / bottom bits: long stretches without a swap, then swaps every bit) over lane-distinct curve points, and
    4 107 random scalars over BASE "points" whose coordinates come from the family (any (x, y) drives the same expression DAG)."""
    from test_oracle import digit_pattern_operands
    fam = digit_pattern_operands()
    k = fam[::100]
    n = len(k)
    bx, by = _lane_distinct_points(gpu, cv, n)
    exp = oracle.scalar_mult(cv, k, bx, by, threads=THREADS)
    assert all(np.array_equal(u, v) for u, v in zip(gpu.scalar_mult(cv, k, bx, by), exp)), "pattern scalars"
    a = fam[::409]
    m = len(a)
    p = np.tile(oracle.constants(cv)["p"], (m, 1))
    x, y = oracle.sub_if_above(a, p), oracle.sub_if_above(np.roll(a, 77, axis=0), p)
    kr = fill_random_np(m, SEED, 21)
    exp = oracle.scalar_mult(cv, kr, x, y, threads=THREADS)
    assert all(np.array_equal(u, v) for u, v in zip(gpu.scalar_mult(cv, kr, x, y), exp)), "pattern coordinates"
    exp = oracle.scalar_mult(cv, k[:m], x, y, threads=THREADS)
    assert all(np.array_equal(u, v) for u, v in zip(gpu.scalar_mult(cv, k[:m], x, y), exp)), "pattern scalars on pattern coordinates"


@pytest.mark.parametrize("cv", CURVES)
def test_windowed_paths_on_digit_pattern_scalars(engine, cv):
    """The window algorithms recode the scalar (signed 4-, 7- and 20-bit digits with carries between windows, odd-digit forms on k or
    n - k, the GLV split with its rounding on secp256k1): 16 796 scalars from the digit-pattern family -- nibbles 0, 7, 8, f in every
    position, runs of ones across window boundaries -- through every one of them and the x-only ladder, against the reference ladder's
    affine result (itself held to the oracle on these scalars by the test above).  None of the family is a degenerate scalar of the
    ladder except k = 0, where both sides give (0, 0)."""
    import torch
    from ecsimd_amd import ALG_NO_ENDOMORPHISM
    from test_oracle import digit_pattern_operands
    kn = digit_pattern_operands()[::100]
    n = len(kn)
    k = engine.to_device(kn)
    s = engine.fill_random(n, SEED, 2)
    bx, by = engine.scalar_mult_base(cv, s, flags=OUT_AFFINE)
    lx, ly = engine.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE)                       # the reference's algorithm
    for alg, name in ((ALG_WINDOWED, "per-element tables"), (ALG_WINDOWED | ALG_NO_ENDOMORPHISM, "per-element tables, plain odd-digit loop")):
        wx, wy = engine.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE | alg)
        assert torch.equal(wx, lx) and torch.equal(wy, ly), name
    xo, none = engine.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE, x_only=True)
    assert none is None and torch.equal(xo, lx), "x-only ladder"
    gx, gy = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE)
    for alg, name in ((ALG_WINDOWED, "4-bit LDS table"), (ALG_WINDOWED_SIGNED, "signed 7-bit LDS table"), (ALG_WINDOWED_BIG, "20-bit table")):
        fx, fy = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | alg)
        assert torch.equal(fx, gx) and torch.equal(fy, gy), name
    # u1*G + u2*Q with both scalars from the family (Q valid public keys): against the two ladders and an affine addition
    u2 = engine.to_device(np.roll(kn, 4321, axis=0))
    px, py = engine.scalar_mult(cv, u2, bx, by, flags=OUT_AFFINE)
    ex, ey, efin = engine.affine_add(cv, (gx, gy), (px, py))
    rx, ry, fin = engine.double_scalar_mult(cv, k, u2, bx, by)
    assert torch.equal(rx, ex) and torch.equal(ry, ey) and torch.equal(fin, efin), "u1*G + u2*Q"


@pytest.mark.parametrize("cv", CURVES)
def test_scalar_mult_vs_oracle(gpu, oracle, cv):
    c = CURVE_PARAMS[cv]; order = c["n"]
    edge = [0, 1, 2, 3, 4, 5, 6, 7, 8, order - 2, order - 1, order, order + 1, order + 2, 2**256 - 1, 2**256 - 2, 2**255, 2**255 - 1,
            2**64, 2**64 - 1, 2**128, 2**192 + 1, int("55" * 32, 16), int("aa" * 32, 16),
            2**256 - order, 2**256 - order - 1, 2**256 - order + 1]      # k = 2^256 mod n (and k + 1): the ladder meets n*P at its last step
    n = 4096
    k = fill_random_np(n, SEED, 1); k[:len(edge)] = ints_to_arr(edge)
    gx, gy = ints_to_arr([c["gx"]] * n), ints_to_arr([c["gy"]] * n)
    exp = oracle.scalar_mult(cv, k, gx, gy, threads=THREADS)
    assert all(np.array_equal(u, v) for u, v in zip(gpu.scalar_mult(cv, k, gx, gy), exp)), "fixed base G"
    assert all(np.array_equal(u, v) for u, v in zip(gpu.scalar_mult_base(cv, k), exp)), "scalar_mult_base"
    # variable (lane-distinct) base, both input forms, both output forms
    bx, by = _lane_distinct_points(gpu, cv, n)
    exp = oracle.scalar_mult(cv, k, bx, by, threads=THREADS)
    assert all(np.array_equal(u, v) for u, v in zip(gpu.scalar_mult(cv, k, bx, by), exp)), "variable base"
    Pm = oracle.from_affine(cv, bx, by)
    assert all(np.array_equal(u, v) for u, v in zip(gpu.scalar_mult(cv, k, Pm[0], Pm[1], mgry_in=True), exp)), "Montgomery-form base"
    ax, ay = oracle.to_affine(cv, exp)
    gax, gay = gpu.scalar_mult(cv, k, bx, by, affine=True)[:2]
    assert np.array_equal(gax, ax) and np.array_equal(gay, ay), "affine output"
    # a few lanes against the independent affine model (non-degenerate scalars only)
    for i in list(range(1, 9)) + [len(edge) + 3, n - 1]:
        kk = to_int(k[i]) % order
        if kk in (0, order - 1, 2**256 - order, 2**256 - order - 1): continue          # the ladder's degenerate scalars (level J only)
        assert (to_int(gax[i]), to_int(gay[i])) == ec_mul(cv, kk, (to_int(bx[i]), to_int(by[i]))), i
    # one scalar for all lanes (curve_group.h:221-251)
    k1 = k[len(edge) + 1]
    exp1 = oracle.scalar_mult(cv, np.tile(k1, (256, 1)), bx[:256], by[:256], threads=THREADS)
    assert all(np.array_equal(u, v) for u, v in zip(gpu.scalar_mult_1s(cv, k1, bx[:256], by[:256]), exp1)), "scalar_mult_1s"


def comb_exceptional_scalars(cv):
    """For an odd-digit comb with B = 2^bits the last mixed addition meets R = T at exactly one odd scalar: k* = n - 2 (n mod B) when the
    windows are summed from the top (the 4-bit LDS kernel), k* = n - 2 (n mod B^(windows - 1)) from the bottom (the 20-bit table); the even
    n - k* reaches it through the k -> n - k flip, and k* + n where that still fits 256 bits through the reduction (k_affine.inc comb_special)."""
    n = CURVE_PARAMS[cv]["n"]
    out = []
    # low_bits per SHIPPED window shape: bits (summed from the top: the 4-bit LDS comb) or bits * (windows - 1) (summed from the bottom):
    # 20-bit device table 20 * 12, signed 7-bit LDS comb 7 * 36, the constant-time 5-bit comb 5 * 51 (= 255: k* = 2^256 - n, one of the
    # LADDER's degenerate scalars -- ADVICE r3), its 6-bit alternative 6 * 42; 20 / 7 / 249 are earlier shapes, kept.
    for low_bits in (4, 20 * 12, 7 * 36, 5 * 51, 6 * 42, 20, 7, 249):
        m = n % (1 << low_bits)
        out += [n - 2 * m, (2 * m) % n, n - 2 * m + 1, n - 2 * m - 1]
    out += [v + n for v in out if v + n < (1 << 256)]
    return [v for v in out if v % n != 0]


@pytest.mark.parametrize("cv", CURVES)
def test_constant_time_fixed_base(engine, oracle, cv):
    """ECSIMD_HIP_ALG_CONSTANT_TIME (scalar_mult_base + ALG_WINDOWED): the same 4-bit comb reading every table entry under lane masks.
    Same (x, y) as the default kernel on 2^18 random scalars, on k = 0, n, 1, n - 1 and ragged lengths; a sample against the oracle's
    ladder; x only; and the flag is refused wherever it would promise something the kernel behind it does not do."""
    import torch
    c = CURVE_PARAMS[cv]; order = c["n"]
    n = (1 << 18) + 37
    k = engine.fill_random(n, SEED, 61)
    edge = engine.to_device(ints_to_arr([0, order, 1, order - 1, 2, order + 1, 2**256 - 1, order - 2]))
    k[:8] = edge
    # the comb's own exceptional scalar k* = 2^256 - n (5-bit windows summed from the bottom), the even 2n - 2^256 that the k -> n - k flip
    # maps onto it, and their neighbours: 2^256 - n is one of the LADDER's degenerate scalars, so the witness is the big-int model
    star = [2**256 - order, 2 * order - 2**256, 2**256 - order - 1, 2**256 - order + 1, 2 * order - 2**256 - 1, 2 * order - 2**256 + 1, 2**256 - order - 2, 2**256 - order + 2]
    k[8:16] = engine.to_device(ints_to_arr(star))
    ct = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME)
    pl = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | ALG_WINDOWED)
    assert torch.equal(ct[0], pl[0]) and torch.equal(ct[1], pl[1])
    assert not ct[0][:2].any() and not ct[1][:2].any()                        # k = 0 mod n: (0, 0)
    for j, v in enumerate(star):
        mx, my = ec_mul(cv, v % order, (c["gx"], c["gy"]))
        assert arr_to_ints(engine.to_numpy(ct[0][8 + j:9 + j]))[0] == mx and arr_to_ints(engine.to_numpy(ct[1][8 + j:9 + j]))[0] == my, hex(v)
    m = 2048
    kk = engine.to_numpy(k[16:16 + m])
    gx, gy = ints_to_arr([c["gx"]] * m), ints_to_arr([c["gy"]] * m)
    ex, ey = oracle.to_affine(cv, oracle.scalar_mult(cv, kk, gx, gy, threads=THREADS))
    assert np.array_equal(engine.to_numpy(ct[0][16:16 + m]), ex) and np.array_equal(engine.to_numpy(ct[1][16:16 + m]), ey)
    xo = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME, x_only=True)
    assert torch.equal(xo[0], ct[0])
    for n_small in (1, 63, 65, 257):
        a = engine.scalar_mult_base(cv, k[:n_small].contiguous(), flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME)
        assert torch.equal(a[0], ct[0][:n_small]) and torch.equal(a[1], ct[1][:n_small])
    for flags in (OUT_AFFINE | ALG_WINDOWED_SIGNED | ALG_CONSTANT_TIME, OUT_AFFINE | ALG_WINDOWED_BIG | ALG_CONSTANT_TIME, OUT_AFFINE | ALG_CONSTANT_TIME):
        with pytest.raises(EcsimdHipError, match="ALG_CONSTANT_TIME"):
            engine.scalar_mult_base(cv, k[:64].contiguous(), flags=flags)
    # the same rule at every entry point: the flag modifies ALG_WINDOWED and nothing else -- not the signed / big tables, not the ladder
    for flags in (OUT_AFFINE | ALG_WINDOWED_SIGNED | ALG_CONSTANT_TIME, OUT_AFFINE | ALG_CONSTANT_TIME, ALG_CONSTANT_TIME):
        with pytest.raises(EcsimdHipError, match="ALG_CONSTANT_TIME"):
            engine.scalar_mult(cv, k[:64].contiguous(), ct[0][:64].contiguous(), ct[1][:64].contiguous(), flags=flags)
        with pytest.raises(EcsimdHipError, match="ALG_CONSTANT_TIME"):
            engine.scalar_mult_1s(cv, from_int(12345), ct[0][:64].contiguous(), ct[1][:64].contiguous(), flags=flags)


@pytest.mark.parametrize("cv", CURVES)
def test_constant_time_variable_base(engine, oracle, cv):
    """ALG_WINDOWED | ALG_CONSTANT_TIME on a variable base: the per-element window tables with EVERY entry of the lane's table read in every
    window (on secp256k1: the GLV split on the complete addition law, and with ALG_NO_ENDOMORPHISM the plain odd-digit loop).  Same (x, y) as the default window loop on 2^17 + 5 lane-distinct (k, P) with the edge scalars
    in front, x only, the shared-scalar form; a sample against the oracle's ladder."""
    import torch
    c = CURVE_PARAMS[cv]; order = c["n"]
    n = (1 << 17) + 5
    k = engine.fill_random(n, SEED, 71)
    k[:8] = engine.to_device(ints_to_arr([0, order, 1, order - 1, 2, order + 1, 2**256 - 1, order - 2]))
    px, py = engine.scalar_mult_base(cv, engine.fill_random(n, SEED, 72), flags=OUT_AFFINE | ALG_WINDOWED_BIG)
    ct = engine.scalar_mult(cv, k, px, py, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME)
    pl = engine.scalar_mult(cv, k, px, py, flags=OUT_AFFINE | ALG_WINDOWED)
    assert torch.equal(ct[0], pl[0]) and torch.equal(ct[1], pl[1])
    assert not ct[0][:2].any() and not ct[1][:2].any()                        # k = 0 mod n: (0, 0)
    m = 1024
    ex, ey = oracle.to_affine(cv, oracle.scalar_mult(cv, engine.to_numpy(k[8:8 + m]), engine.to_numpy(px[8:8 + m]), engine.to_numpy(py[8:8 + m]), threads=THREADS))
    assert np.array_equal(engine.to_numpy(ct[0][8:8 + m]), ex) and np.array_equal(engine.to_numpy(ct[1][8:8 + m]), ey)
    xo = engine.scalar_mult(cv, k, px, py, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME, x_only=True)
    assert torch.equal(xo[0], ct[0])
    ne = engine.scalar_mult(cv, k, px, py, flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME | ALG_NO_ENDOMORPHISM)     # secp256k1: the other loop
    assert torch.equal(ne[0], pl[0]) and torch.equal(ne[1], pl[1])
    k1 = engine.to_numpy(k[100:101])[0]
    a = engine.scalar_mult_1s(cv, k1, px[:4096].contiguous(), py[:4096].contiguous(), flags=OUT_AFFINE | ALG_WINDOWED | ALG_CONSTANT_TIME)
    b = engine.scalar_mult_1s(cv, k1, px[:4096].contiguous(), py[:4096].contiguous(), flags=OUT_AFFINE | ALG_WINDOWED)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("cv", CURVES)
def test_comb_kernels_on_their_exceptional_scalars(engine, cv):
    """Round 3 found k = 2 wrong in the odd-digit 4-bit kernel on P-256 (R = T at its last addition) and, by the same argument, k = +-(n - 2
    (n mod 2^240)) wrong in the 20-bit kernel on BOTH curves since round 1 -- scalars no random test meets.  The kernels now substitute
    the point of that one scalar (built with the table); every fixed-base algorithm and u1*G + u2*Q must give the big-int model's point on
    them, on their neighbours, in waves where every / one / no lane is such a scalar."""
    import torch
    c = CURVE_PARAMS[cv]; order = c["n"]; G = (c["gx"], c["gy"])
    sp = comb_exceptional_scalars(cv)
    rnd = arr_to_ints(fill_random_np(192, SEED, 33))
    ks = sp * 3 + rnd[:64] + [sp[0]] + rnd[64:127] + rnd[127:]              # waves full of them, one wave with a single one, one without
    k = engine.to_device(ints_to_arr(ks))
    model = {v: ec_mul(cv, v % order, G) for v in set(ks)}                    # the independent big-int model is the witness: n - 1, one of the
    exp = np.array([[from_int(model[v][0]), from_int(model[v][1])] for v in ks])   # ladder's own degenerate scalars, is a neighbour of k* = n - 2
    ex_, ey_ = engine.to_device(exp[:, 0]), engine.to_device(exp[:, 1])
    degenerate = {order - 1, 2**256 - order - 1, 2**256 - order}
    lx, ly = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE)
    assert all(torch.equal(lx[i], ex_[i]) and torch.equal(ly[i], ey_[i]) for i in range(len(ks)) if ks[i] not in degenerate), "ladder"
    for alg, name in ((ALG_WINDOWED, "4-bit LDS table"), (ALG_WINDOWED | ALG_CONSTANT_TIME, "5-bit LDS comb, every entry read (constant time)"),
                      (ALG_WINDOWED_SIGNED, "signed 7-bit LDS table"), (ALG_WINDOWED_BIG, "20-bit table")):
        wx, wy = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | alg)
        bad = [hex(ks[i]) for i in range(len(ks)) if not (torch.equal(wx[i], ex_[i]) and torch.equal(wy[i], ey_[i]))]
        assert not bad, (name, bad)
    lx, ly = ex_, ey_
    # u1*G + u2*Q with u1 from the list (the 20-bit table now exists in this context, so u1*G goes through it)
    u2 = engine.fill_random(len(ks), SEED, 34)
    qx, qy = engine.scalar_mult_base(cv, engine.fill_random(len(ks), SEED, 35), flags=OUT_AFFINE)
    px, py = engine.scalar_mult(cv, u2, qx, qy, flags=OUT_AFFINE)
    ex, ey, efin = engine.affine_add(cv, (lx, ly), (px, py))
    rx, ry, fin = engine.double_scalar_mult(cv, k, u2, qx, qy)
    assert torch.equal(rx, ex) and torch.equal(ry, ey) and torch.equal(fin, efin)
    # variable base: the per-element window loops (Horner form: no such scalar, checked all the same)
    for alg in (ALG_WINDOWED,):
        vx, vy = engine.scalar_mult(cv, k, qx, qy, flags=OUT_AFFINE | alg)
        wx, wy = engine.scalar_mult(cv, k, qx, qy, flags=OUT_AFFINE)
        ok = [i for i in range(len(ks)) if ks[i] not in degenerate]
        assert all(torch.equal(vx[i], wx[i]) and torch.equal(vy[i], wy[i]) for i in ok)


@pytest.mark.parametrize("cv", CURVES)
def test_windowed_fixed_base_matches_the_ladder_at_affine_level(engine, oracle, cv):
    """BASELINE configs[2] algorithm: 4-bit windows over an LDS table + simultaneous inversion.  A different
    algorithm from the reference's ladder, so parity is affine-level (SURVEY.md 8(a) level A): identical
    (x, y) for every non-degenerate scalar; k = 0 mod n gives (0, 0)."""
    import torch
    from ecsimd_amd import OUT_AFFINE, ALG_WINDOWED
    c = CURVE_PARAMS[cv]; order = c["n"]
    edge = [1, 2, 3, 15, 16, 17, 255, 256, 2**64 - 1, 2**64, 2**128 + 1, 2**252, 15 * 2**252, order - 2, order + 1, order + 2, 2**256 - 1,
            0x7fff, 0x8000, 0x8001, 0xffff, 0x10000, 0x18000, int("8000" * 16, 16), int("7fff" * 16, 16), int("8001" * 16, 16),   # 16-bit window boundaries (an earlier table width)
            0x7ffff, 0x80000, 0x80001, 0xfffff, 0x100000, 0x180000, int("80000" * 12, 16), int("7ffff" * 12, 16), int("80001" * 12, 16),   # 20-bit window boundaries
            int("f0" * 32, 16), int("0f" * 32, 16), int("10" * 32, 16), 2**255, 0x1000000000000000000000000000000000000000000000000000000000000000]
    n = (1 << 18) + 77                                      # ragged, and large enough for several elements per lane in the inversion
    k = fill_random_np(n, SEED, 5); k[:len(edge)] = ints_to_arr(edge)
    k[100] = from_int(0); k[101] = from_int(order); k[102] = from_int(order - 1)       # degenerate scalars
    k[103] = from_int(2**256 - order); k[104] = from_int(2**256 - order - 1)           # ... of the ladder only
    dk = engine.to_device(k)
    wx, wy = engine.scalar_mult_base(cv, dk, flags=OUT_AFFINE | ALG_WINDOWED)
    from ecsimd_amd import ALG_WINDOWED_SIGNED
    w6x, w6y = engine.scalar_mult_base(cv, dk, flags=OUT_AFFINE | ALG_WINDOWED_SIGNED)       # signed 7-bit windows: same points
    assert torch.equal(w6x, wx) and torch.equal(w6y, wy)
    from ecsimd_amd import ALG_WINDOWED_BIG
    wbx, wby = engine.scalar_mult_base(cv, dk, flags=OUT_AFFINE | ALG_WINDOWED_BIG)          # signed 20-bit windows, table in device memory
    assert torch.equal(wbx, wx) and torch.equal(wby, wy)
    lx, ly = engine.scalar_mult_base(cv, dk, flags=OUT_AFFINE)                        # reference ladder + (batched) to_affine
    wxn, wyn, lxn, lyn = (engine.to_numpy(t) for t in (wx, wy, lx, ly))
    keep = np.ones(n, dtype=bool); keep[100:105] = False
    assert np.array_equal(wxn[keep], lxn[keep]) and np.array_equal(wyn[keep], lyn[keep])
    assert to_int(wxn[100]) == 0 and to_int(wyn[100]) == 0 and to_int(wxn[101]) == 0 and to_int(wyn[101]) == 0
    G = (c["gx"], c["gy"])
    assert (to_int(wxn[102]), to_int(wyn[102])) == (c["gx"], c["p"] - c["gy"])                  # (n-1)G = -G: the windowed path is right where the ladder degenerates
    for i in list(range(len(edge))) + [103, 104, n - 1]:                 # 103/104: right where the reference ladder is wrong
        assert (to_int(wxn[i]), to_int(wyn[i])) == ec_mul(cv, to_int(k[i]) % order, G), hex(to_int(k[i]))
    m = 2048
    ex, ey = oracle.to_affine(cv, oracle.scalar_mult(cv, k[200:200 + m], ints_to_arr([c["gx"]] * m), ints_to_arr([c["gy"]] * m), threads=THREADS))
    assert np.array_equal(wxn[200:200 + m], ex) and np.array_equal(wyn[200:200 + m], ey)
    import ctypes as C
    assert engine.lib.ecsimd_hip_scalar_mult_base(engine.ctx, C.c_int(cv), C.c_void_p(dk.data_ptr()), C.c_void_p(wx.data_ptr()), C.c_void_p(wy.data_ptr()),
                                                   C.c_void_p(wx.data_ptr()), C.c_size_t(4), C.c_int(ALG_WINDOWED)) == -1      # Jacobian out is not offered


@pytest.mark.parametrize("cv", CURVES)
def test_windowed_variable_base_matches_the_ladder_at_affine_level(engine, oracle, cv):
    """ecsimd_hip_scalar_mult with ALG_WINDOWED: per-lane tables {1..8}P + signed 4-bit windows.
    A different algorithm from the reference's ladder, so parity is affine-level: identical (x, y) to the ladder +
    to_affine for every non-degenerate scalar, the big-int model on the scalars where the ladder degenerates,
    (0, 0) for k = 0 mod n; classical and Montgomery-form base points; more lanes than one internal chunk."""
    import torch
    import ctypes as C
    from ecsimd_amd import OUT_AFFINE, ALG_WINDOWED, BASE_MGRY
    c = CURVE_PARAMS[cv]; order = c["n"]
    edge = [0, order, 1, 2, 7, 8, 9, 15, 16, 17, 0x78, 0x80, 0x88, 2**252, 2**255, (order - 1) // 2, (order + 1) // 2, order - 2, order - 1,
            order + 1, order + 9, 2**256 - 1, 2**256 - order, 2**256 - order - 1, int("8" * 64, 16), int("7" * 64, 16), int("9" * 64, 16),
            int("08" * 32, 16), int("80" * 32, 16), int("f0" * 32, 16), int("0f" * 32, 16)]
    # scalars around the GLV lattice of secp256k1 (lambda, the basis vectors, their neighbours): the split's corner cases
    lam = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
    a1, mb1, a2 = 0x3086d221a7d46bcde86c90e49284eb15, 0xe4437ed6010e88286f547fa90abfe4c3, 0x114ca50f7a8e2f3f657c1108d9d44cfd8
    edge += [lam, lam + 1, lam - 1, (2 * lam) % order, order - lam, (lam * lam) % order, a1, a1 + 1, a1 - 1, mb1, mb1 + 8, a2, a2 - 8, (a1 * lam) % order,
             (mb1 * lam) % order, (a1 + mb1 * lam) % order, (8 + 8 * lam) % order, (order - 8 - 8 * lam) % order, (1 << 128) - 1, 1 << 128, ((1 << 128) * lam) % order,
             (((1 << 128) - 1) * (lam + 1)) % order, (0x88888888888888888888888888888888 * (lam + 1)) % order]
    n = (1 << 22) + 4099 if cv == P256 else (1 << 18) + 77          # P-256: crosses the 2^22-lane chunk boundary
    k = engine.fill_random(n, SEED, 53); s = engine.fill_random(n, SEED, 54)
    k[:len(edge)] = engine.to_device(ints_to_arr(edge))
    bx, by = engine.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED)
    wx, wy = engine.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    lx, ly = engine.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE)                   # reference ladder + batched to_affine
    m = len(edge)
    assert torch.equal(wx[m:], lx[m:]) and torch.equal(wy[m:], ly[m:])
    wxn, wyn, bxn, byn = (engine.to_numpy(t[:m]) for t in (wx, wy, bx, by))
    for i, kv in enumerate(edge):
        exp = ec_mul(cv, kv % order, (to_int(bxn[i]), to_int(byn[i]))) or (0, 0)
        assert (to_int(wxn[i]), to_int(wyn[i])) == exp, hex(kv)
    P = engine.from_affine(cv, bx[:8192].contiguous(), by[:8192].contiguous())   # Montgomery-form base point
    mx, my = engine.scalar_mult(cv, k[:8192].contiguous(), P[0], P[1], flags=OUT_AFFINE | ALG_WINDOWED | BASE_MGRY)
    assert torch.equal(mx, wx[:8192]) and torch.equal(my, wy[:8192])
    ix, iy = bx[:8192].clone(), by[:8192].clone()                                # in place: outputs over the base point
    engine.scalar_mult(cv, k[:8192].contiguous(), ix, iy, flags=OUT_AFFINE | ALG_WINDOWED, out=[ix, iy, None])
    assert torch.equal(ix, wx[:8192]) and torch.equal(iy, wy[:8192])
    from ecsimd_amd import ALG_NO_ENDOMORPHISM                                   # secp256k1: GLV split by default, the plain loop on request
    qx, qy = engine.scalar_mult(cv, k[:1 << 17].contiguous(), bx[:1 << 17].contiguous(), by[:1 << 17].contiguous(), flags=OUT_AFFINE | ALG_WINDOWED | ALG_NO_ENDOMORPHISM)
    assert torch.equal(qx, wx[:1 << 17]) and torch.equal(qy, wy[:1 << 17])
    k1 = engine.to_numpy(k[m + 5])                                               # one scalar for every lane (scalar_mult_1s)
    sx, sy = engine.scalar_mult_1s(cv, k1, bx[:8192].contiguous(), by[:8192].contiguous(), flags=OUT_AFFINE | ALG_WINDOWED)
    tx, ty = engine.scalar_mult_1s(cv, k1, bx[:8192].contiguous(), by[:8192].contiguous(), flags=OUT_AFFINE)
    assert torch.equal(sx, tx) and torch.equal(sy, ty)
    j = 4096
    ex, ey = oracle.to_affine(cv, oracle.scalar_mult(cv, engine.to_numpy(k[m:m + j]), engine.to_numpy(bx[m:m + j]), engine.to_numpy(by[m:m + j]), threads=THREADS))
    assert np.array_equal(engine.to_numpy(wx[m:m + j]), ex) and np.array_equal(engine.to_numpy(wy[m:m + j]), ey)
    p_ = lambda t: C.c_void_p(t.data_ptr())
    assert engine.lib.ecsimd_hip_scalar_mult(engine.ctx, C.c_int(cv), p_(k), p_(bx), p_(by), p_(wx), p_(wy), p_(wx), C.c_size_t(4), C.c_int(ALG_WINDOWED)) == -1   # Jacobian out is not offered


@pytest.mark.parametrize("cv", CURVES)
def test_invalid_base_points_do_not_poison_their_neighbours_in_the_windowed_paths(engine, cv):
    """scalar_mult does not validate its base points (the reference does not either), but the windowed paths SHARE inversions between lanes --
    the table step (one inversion per lane's chain of multiples, k_varwin_invert_last, taken over many lanes at once) and the final to_affine --
    so a lane whose "point" is (0, 0), off the curve, or has a coordinate >= p must cost its neighbours nothing: every other lane equals the ladder,
    in every windowed form (default, constant time, the plain loop on secp256k1), for small and chunk-sized batches."""
    import torch
    from ecsimd_amd import OUT_AFFINE, ALG_WINDOWED, ALG_CONSTANT_TIME, ALG_NO_ENDOMORPHISM
    for n in (5, 300, (1 << 18) + 3):
        k = engine.fill_random(n, SEED, 91); s = engine.fill_random(n, SEED, 92)
        bx, by = engine.scalar_mult_base(cv, s, flags=OUT_AFFINE)
        lx, ly = engine.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE)
        bad = sorted({0, 3, n - 1, n // 2, min(n - 1, 64), min(n - 1, 255), min(n - 1, 256)})
        px, py = bx.clone(), by.clone()
        for j, i in enumerate(bad):
            if j % 3 == 0:   px[i] = 0; py[i] = 0                                  # "infinity"
            elif j % 3 == 1: py[i] = px[i]                                         # off the curve
            else:            px[i] = -1; py[i] = -1                                # 2^256 - 1 >= p
        good = torch.ones(n, dtype=torch.bool, device=bx.device); good[bad] = False
        for flags in (ALG_WINDOWED, ALG_WINDOWED | ALG_CONSTANT_TIME, ALG_WINDOWED | ALG_NO_ENDOMORPHISM):
            wx, wy = engine.scalar_mult(cv, k, px, py, flags=OUT_AFFINE | flags)
            assert torch.equal(wx[good], lx[good]) and torch.equal(wy[good], ly[good]), (n, flags)


@pytest.mark.parametrize("cv", CURVES)
def test_simultaneous_inversion_to_affine(engine, oracle, cv):
    """to_affine with Montgomery's trick (non-aliasing outputs) == one inversion per element (aliasing
    outputs force the per-element kernel) == the oracle; Z = 0 elements give (0, 0) and do not poison
    their neighbours."""
    import torch
    n = (1 << 19) + 5
    k = engine.fill_random(n, SEED, 6); s = engine.fill_random(n, SEED, 2)
    bx, by = engine.scalar_mult_base(cv, s, flags=2)
    J = [t.clone() for t in engine.scalar_mult(cv, k, bx, by)]
    for i in (0, 1, 77, n - 1, n // 2):
        J[2][i] = 0                                                             # Z = 0
    ax, ay = engine.to_affine(cv, J)                                            # batched
    Jc = [t.clone() for t in J]
    import ctypes as C
    p = lambda t: C.c_void_p(t.data_ptr())
    assert engine.lib.ecsimd_hip_to_affine(engine.ctx, C.c_int(cv), p(Jc[0]), p(Jc[1]), p(Jc[2]), p(Jc[0]), p(Jc[1]), C.c_size_t(n)) == 0   # in place
    assert torch.equal(ax, Jc[0]) and torch.equal(ay, Jc[1])
    for i in (0, 1, 77, n - 1, n // 2):
        assert int(ax[i].abs().sum()) == 0 and int(ay[i].abs().sum()) == 0
    idx = np.concatenate([np.arange(0, 64), np.arange(n - 64, n), np.arange(0, n, n // 1024)])
    tidx = torch.from_numpy(idx).to(ax.device)
    ex, ey = oracle.to_affine(cv, tuple(engine.to_numpy(t[tidx]) for t in J))
    assert np.array_equal(engine.to_numpy(ax[tidx]), ex) and np.array_equal(engine.to_numpy(ay[tidx]), ey)


@pytest.mark.parametrize("cv", CURVES)
def test_affine_add_and_double_scalar_mult(engine, oracle, cv):
    """u1*G + u2*Q (the ECDSA-verification shape) = windowed fixed base + reference ladder + one batched affine
    addition, against the independent affine big-int model; the addition's special cases (doubling, inverse
    points, infinity operands) one by one."""
    import torch
    c = CURVE_PARAMS[cv]; order = c["n"]; p = c["p"]; G = (c["gx"], c["gy"])
    n = (1 << 17) + 33
    u1 = fill_random_np(n, SEED, 21); u2 = fill_random_np(n, SEED, 22); s = fill_random_np(n, SEED, 2)
    u1[0] = from_int(0); u2[1] = from_int(1); u1[2] = from_int(order)
    qx, qy = engine.scalar_mult_base(cv, engine.to_device(s), flags=2)
    rx, ry, fin = engine.double_scalar_mult(cv, engine.to_device(u1), engine.to_device(u2), qx, qy)
    rxn, ryn, qxn, qyn = (engine.to_numpy(t) for t in (rx, ry, qx, qy))
    assert bool(fin.all())
    for i in list(range(8)) + [n // 2, n - 1]:
        Q = (to_int(qxn[i]), to_int(qyn[i]))
        exp = ec_add(cv, ec_mul(cv, to_int(u1[i]) % order, G), ec_mul(cv, to_int(u2[i]) % order, Q))
        assert (to_int(rxn[i]), to_int(ryn[i])) == exp, i
    xo, _, _ = engine.double_scalar_mult(cv, engine.to_device(u1), engine.to_device(u2), qx, qy, x_only=True)
    assert torch.equal(xo, rx)
    # consistency on the whole batch: the same sum from the two affine products and affine_add
    gx_, gy_ = engine.scalar_mult_base(cv, engine.to_device(u1), flags=6)
    px_, py_ = engine.scalar_mult(cv, engine.to_device(u2), qx, qy, flags=2)
    sx, sy, sf = engine.affine_add(cv, (gx_, gy_), (px_, py_))
    assert torch.equal(sx, rx) and torch.equal(sy, ry) and bool(sf.all())
    # special cases of the addition
    P1 = ec_mul(cv, 5, G); P2 = ec_mul(cv, 7, G); negP1 = (P1[0], p - P1[1])
    A = [P1, P1, P1, (0, 0), P1, (0, 0)]
    B = [P2, P1, negP1, P2, (0, 0), (0, 0)]
    E = [ec_add(cv, P1, P2), ec_add(cv, P1, P1), None, P2, P1, None]
    ax = engine.to_device(ints_to_arr([a[0] for a in A])); ay = engine.to_device(ints_to_arr([a[1] for a in A]))
    bx = engine.to_device(ints_to_arr([b[0] for b in B])); by = engine.to_device(ints_to_arr([b[1] for b in B]))
    tx, ty, tf = engine.affine_add(cv, (ax, ay), (bx, by))
    txn, tyn, tfn = engine.to_numpy(tx), engine.to_numpy(ty), engine.to_numpy(tf)
    for i, e_ in enumerate(E):
        if e_ is None:
            assert tfn[i] == 0 and to_int(txn[i]) == 0 and to_int(tyn[i]) == 0
        else:
            assert tfn[i] == 1 and (to_int(txn[i]), to_int(tyn[i])) == e_, i


@pytest.mark.parametrize("cv", CURVES)
def test_ecdsa_acceptance_test(engine, openssl, cv):
    """ecdsa_verify_rx on real signatures: d random, Q = d*G, k random, r = (k*G).x mod n, s = (e + r d)/k, u1 = e/s,
    u2 = r/s (host big-int arithmetic mod n; the points from OpenSSL).  Valid ones pass; a changed r, e or Q fails."""
    c = CURVE_PARAMS[cv]; order = c["n"]
    n = 2048
    rng = np.random.default_rng(12345 + cv)
    rand = lambda: [int.from_bytes(rng.bytes(32), "big") % (order - 1) + 1 for _ in range(n)]
    d, kk, e_ = rand(), rand(), rand()
    qx, qy, _ = openssl.scalar_mult_base(cv, ints_to_arr(d), threads=THREADS)
    kx, _, _ = openssl.scalar_mult_base(cv, ints_to_arr(kk), threads=THREADS)
    r = [to_int(v) % order for v in kx]
    s_ = [(e_[i] + r[i] * d[i]) * pow(kk[i], -1, order) % order for i in range(n)]
    good = [i for i in range(n) if r[i] and s_[i]]
    w = [pow(s_[i], -1, order) if s_[i] else 1 for i in range(n)]
    u1 = [e_[i] * w[i] % order for i in range(n)]; u2 = [r[i] * w[i] % order for i in range(n)]
    dev = lambda v: engine.to_device(ints_to_arr(v))
    ok = engine.to_numpy(engine.ecdsa_verify_rx(cv, dev(u1), dev(u2), engine.to_device(qx), engine.to_device(qy), dev(r)))
    assert all(ok[i] == 1 for i in good) and len(good) > n - 4
    bad_r = [(v + 1) % order for v in r]
    assert not engine.to_numpy(engine.ecdsa_verify_rx(cv, dev(u1), dev(u2), engine.to_device(qx), engine.to_device(qy), dev(bad_r))).any()
    u1b = [(e_[i] + 1) * w[i] % order for i in range(n)]                               # another message
    assert not engine.to_numpy(engine.ecdsa_verify_rx(cv, dev(u1b), dev(u2), engine.to_device(qx), engine.to_device(qy), dev(r))).any()
    qx2 = np.roll(qx, 1, axis=0); qy2 = np.roll(qy, 1, axis=0)                         # somebody else's key
    assert not engine.to_numpy(engine.ecdsa_verify_rx(cv, dev(u1), dev(u2), engine.to_device(qx2), engine.to_device(qy2), dev(r))).any()


@pytest.mark.parametrize("cv", CURVES)
def test_small_base_batches_take_the_comb_and_keep_the_ladders_bits(engine, cv):
    """scalar_mult_base with OUT_AFFINE and no algorithm flag: up to 2^16 lanes go through the constant-time comb (0.2 ms instead of a 1.3 ms
    ladder launch) and must return the LADDER's affine bits -- at its three degenerate scalars too, where the ladder's point is not k*G
    (the lanes take the ladder's coordinates from the context's record).  Checked against the ladder itself run on G as a variable base,
    on every edge scalar, at the sizes around the route's limit, x-only included, and through ecsimd_hip_scalar_mult with x = y = NULL | BASE_GENERATOR."""
    import torch
    from ecsimd_amd import OUT_AFFINE, LADDER_RADIX32, BASE_GENERATOR
    c = CURVE_PARAMS[cv]; order = c["n"]
    edge = [0, 1, 2, 3, order - 2, order - 1, order, order + 1, 2**256 - order - 2, 2**256 - order - 1, 2**256 - order, 2**256 - order + 1,
            2**256 - 1, 2**255, 2**255 - 1, (order - 1) // 2, (order + 1) // 2, 2 * order - 2**256, 31, 32, 2**5 - 1, 2**250]
    for n in (1, 4, 100, 4096, 1 << 16, (1 << 16) + 1):
        k = engine.fill_random(n, SEED, 70 + cv)
        m = min(n, len(edge))
        k[:m] = engine.to_device(ints_to_arr(edge[:m]))
        gx = engine.to_device(np.tile(from_int(c["gx"]), (n, 1))); gy = engine.to_device(np.tile(from_int(c["gy"]), (n, 1)))
        lx, ly = engine.scalar_mult(cv, k, gx, gy, flags=OUT_AFFINE)                  # the reference's ladder, G as a variable base
        bx, by = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE)                     # the route under test (the ladder itself above 2^16)
        assert torch.equal(bx, lx) and torch.equal(by, ly), (n, np.flatnonzero((engine.to_numpy(bx) != engine.to_numpy(lx)).any(axis=1))[:8])
        fx, fy = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | LADDER_RADIX32)    # an explicit ladder flag keeps the ladder
        assert torch.equal(fx, lx) and torch.equal(fy, ly)
        xo, none = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE, x_only=True)
        xl, _ = engine.scalar_mult(cv, k, gx, gy, flags=OUT_AFFINE, x_only=True)
        assert none is None and torch.equal(xo, xl), n
        if n <= 4096:                                                                 # no base point = the generator, through the variable-base entry point
            ox, oy = engine.empty(n), engine.empty(n)
            engine._bind_stream()
            args = (engine.ctx, C.c_int(cv), C.c_void_p(k.data_ptr()), None, None, C.c_void_p(ox.data_ptr()), C.c_void_p(oy.data_ptr()), None, C.c_size_t(n))
            rc = engine.lib.ecsimd_hip_scalar_mult(*args, C.c_int(OUT_AFFINE | BASE_GENERATOR))
            assert rc == 0 and torch.equal(ox, lx) and torch.equal(oy, ly)
            # two null pointers WITHOUT the flag are an error like any other null pointer (a failed allocation must not yield k G)
            assert engine.lib.ecsimd_hip_scalar_mult(*args, C.c_int(OUT_AFFINE)) == -1
            assert engine.lib.ecsimd_hip_scalar_mult(engine.ctx, C.c_int(cv), C.c_void_p(k.data_ptr()), C.c_void_p(gx.data_ptr()), C.c_void_p(gy.data_ptr()),
                                                     C.c_void_p(ox.data_ptr()), C.c_void_p(oy.data_ptr()), None, C.c_size_t(n), C.c_int(OUT_AFFINE | BASE_GENERATOR)) == -1
    # the Jacobian form is the ladder's at every size (a comb has another representative): level J against the oracle elsewhere; here: unchanged by the route
    k = engine.fill_random(64, SEED, 72)
    gx = engine.to_device(np.tile(from_int(c["gx"]), (64, 1))); gy = engine.to_device(np.tile(from_int(c["gy"]), (64, 1)))
    assert all(torch.equal(a, b) for a, b in zip(engine.scalar_mult_base(cv, k), engine.scalar_mult(cv, k, gx, gy)))


def test_double_scalar_mult_across_the_chunk_boundary(engine):
    """More elements than one internal chunk (2^22): the composite equals its three parts computed separately."""
    import torch
    from ecsimd_amd import OUT_AFFINE, ALG_WINDOWED, ALG_WINDOWED_BIG
    n = (1 << 22) + 777
    u1 = engine.fill_random(n, SEED, 95); u2 = engine.fill_random(n, SEED, 96)
    qx, qy = engine.scalar_mult_base(P256, engine.fill_random(n, SEED, 97), flags=OUT_AFFINE | ALG_WINDOWED_BIG)
    rx, ry, fin = engine.double_scalar_mult(P256, u1, u2, qx, qy)
    g = engine.scalar_mult_base(P256, u1, flags=OUT_AFFINE | ALG_WINDOWED_BIG)
    p_ = engine.scalar_mult(P256, u2, qx, qy, flags=OUT_AFFINE | ALG_WINDOWED)
    sx, sy, sf = engine.affine_add(P256, g, p_)
    assert torch.equal(rx, sx) and torch.equal(ry, sy) and torch.equal(fin, sf) and bool(fin.all())


def test_switching_streams_keeps_the_context_scratch_ordered(engine):
    """Two workspace-using calls back to back on two torch streams: the second must not start on the shared
    scratch (per-element tables, Jacobian intermediates) before the first has finished with it."""
    import torch
    from ecsimd_amd import OUT_AFFINE, ALG_WINDOWED
    n = 1 << 19
    ka, kb = engine.fill_random(n, SEED, 61), engine.fill_random(n, SEED, 62)
    bx, by = engine.scalar_mult_base(P256, engine.fill_random(n, SEED, 63), flags=OUT_AFFINE | ALG_WINDOWED)
    ea = engine.scalar_mult(P256, ka, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    eb = engine.scalar_mult(P256, kb, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(sa):
            ga = engine.scalar_mult(P256, ka, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
        with torch.cuda.stream(sb):
            gb = engine.scalar_mult(P256, kb, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
        torch.cuda.synchronize()
        assert torch.equal(ga[0], ea[0]) and torch.equal(ga[1], ea[1])
        assert torch.equal(gb[0], eb[0]) and torch.equal(gb[1], eb[1])


@pytest.mark.parametrize("cv", CURVES)
def test_complete_mixed_addition(engine, cv):
    """ecsimd_hip_add_mixed_complete: the checked addition of the windowed kernels, every case -- generic, A = B
    (tangent), A = -B, either operand at infinity -- against the affine big-int model."""
    import torch
    c = CURVE_PARAMS[cv]; p = c["p"]; order = c["n"]; G = (c["gx"], c["gy"])
    n = 1024
    ka = fill_random_np(n, SEED, 71); kb = fill_random_np(n, SEED, 72)
    kb[:200] = ka[:200]                                                     # A = B
    kb[200:400] = ints_to_arr([(order - to_int(v) % order) % order for v in ka[200:400]])   # A = -B
    gx = engine.to_device(ints_to_arr([c["gx"]] * n)); gy = engine.to_device(ints_to_arr([c["gy"]] * n))
    A = [t.clone() for t in engine.scalar_mult(cv, engine.to_device(ka), gx, gy)]            # Jacobian, Z != mgry(1)
    bxa, bya = engine.scalar_mult(cv, engine.to_device(kb), gx, gy, flags=2)                 # affine classical
    B = engine.from_affine(cv, bxa, bya)                                                     # Montgomery-form affine in B[0], B[1]
    bx, by = B[0].clone(), B[1].clone()
    for t in A: t[400:500] = 0                                              # A = infinity
    bx[450:600] = 0; by[450:600] = 0                                        # B = infinity (450..499: both)
    R = engine.add_mixed_complete(cv, A, (bx, by))
    rx, ry = engine.to_affine(cv, R)
    rxn, ryn, rzn = engine.to_numpy(rx), engine.to_numpy(ry), engine.to_numpy(R[2])
    for i in list(range(0, 8)) + list(range(196, 204)) + list(range(396, 404)) + list(range(446, 454)) + list(range(496, 504)) + list(range(596, 604)) + [n - 1]:
        Pa = None if 400 <= i < 500 else ec_mul(cv, to_int(ka[i]) % order, G)
        Pb = None if 450 <= i < 600 else ec_mul(cv, to_int(kb[i]) % order, G)
        exp = ec_add(cv, Pa, Pb)
        if exp is None:
            assert to_int(rzn[i]) == 0, i
        else:
            assert (to_int(rxn[i]), to_int(ryn[i])) == exp, i
    # bulk consistency: the generic lanes agree with the reference formula ADD_Z2_1 at the affine level
    Z = engine.add_z2_1(cv, A, (bx, by))
    zx, zy = engine.to_affine(cv, Z)
    assert torch.equal(zx[600:], rx[600:]) and torch.equal(zy[600:], ry[600:])


def test_entry_points_replay_from_a_hip_graph(engine):
    """Capture the ladder, the windowed variable-base path and u1*G + u2*Q into one graph (after a warm-up call that
    sizes the workspace and builds the tables), change the inputs in place, replay: same results as eager calls."""
    import torch
    from ecsimd_amd import OUT_AFFINE, ALG_WINDOWED
    n = 1 << 16
    k = engine.fill_random(n, SEED, 91); s = engine.fill_random(n, SEED, 92)
    bx, by = engine.scalar_mult_base(P256, s, flags=OUT_AFFINE | ALG_WINDOWED)
    J = [engine.empty(n) for _ in range(3)]; W = [engine.empty(n) for _ in range(2)]; D = None

    def run():
        engine.scalar_mult(P256, k, bx, by, out=J)
        engine.scalar_mult(P256, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED, out=W + [None])
        return engine.double_scalar_mult(P256, s, k, bx, by)
    run(); torch.cuda.synchronize()                                   # warm-up: workspace, tables
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            D = run()
    torch.cuda.synchronize()
    k.copy_(engine.fill_random(n, SEED, 93))                          # new scalars, same buffers
    g.replay(); torch.cuda.synchronize()
    gj = [t.clone() for t in J]; gw = [t.clone() for t in W]; gd = [t.clone() for t in D]
    ej = engine.scalar_mult(P256, k, bx, by)
    ew = engine.scalar_mult(P256, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    ed = engine.double_scalar_mult(P256, s, k, bx, by)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(gj, ej))
    assert all(torch.equal(a, b) for a, b in zip(gw, ew))
    assert all(torch.equal(a, b) for a, b in zip(gd, ed))


def test_scalar_mult_p256_entry_point(engine, oracle):
    """lib/scalar_mult_p256.cpp:10-12: scalar_mult_p256(x, P), P Jacobian Montgomery with Z = mgry(1)."""
    n = 1024; c = CURVE_PARAMS[P256]
    k = fill_random_np(n, SEED, 7)
    s = fill_random_np(n, SEED, 8)
    bx, by = engine.scalar_mult_base(P256, engine.to_device(s), flags=2)
    P = engine.from_affine(P256, bx, by)
    got = engine.scalar_mult_p256(engine.to_device(k), P[0], P[1])
    exp = oracle.scalar_mult(P256, k, engine.to_numpy(P[0]), engine.to_numpy(P[1]), threads=THREADS, mgry_in=True)
    assert all(np.array_equal(engine.to_numpy(u), v) for u, v in zip(got, exp))


def test_host_array_form_overlaps_copies_and_ladders_and_returns_the_same_bits(engine):
    """ecsimd_hip_scalar_mult_host (r5): every array in pageable host memory, chunks of 2^19 elements alternating between the context and its helper context.
    The same bits as the device-resident entry point on one element, a ragged small batch, one chunk + 5 and three chunks + 77 (both sides used twice, a ragged
    tail): Jacobian, affine, x only, the generator form, a windowed algorithm on secp256k1, a registered curve, the context's reference-square option on both
    sides; and what it refuses."""
    import torch
    from ecsimd_amd import OUT_AFFINE, ALG_WINDOWED, BASE_GENERATOR, EcsimdHipError
    from ecsimd_amd.curves import curve_id
    big = 3 * (1 << 19) + 77
    k = engine.fill_random(big, SEED, 301); s = engine.fill_random(big, SEED, 302)
    for cv in (P256, SECP256K1, curve_id("brainpoolP256r1")):
        bx, by = engine.scalar_mult_base(cv, s, flags=OUT_AFFINE)
        kn, xn, yn = (engine.to_numpy(t) for t in (k, bx, by))
        for n in ((1, 1000, (1 << 19) + 5, big) if cv == P256 else ((1 << 19) + 5,)):
            dev = [t[:n].contiguous() for t in (k, bx, by)]
            J = engine.scalar_mult(cv, *dev)
            H = engine.scalar_mult_host(cv, kn[:n], xn[:n], yn[:n])
            assert len(H) == 3 and all(np.array_equal(h, engine.to_numpy(j)) for h, j in zip(H, J)), (cv, n)
            A = engine.scalar_mult(cv, *dev, flags=OUT_AFFINE)
            H = engine.scalar_mult_host(cv, kn[:n], xn[:n], yn[:n], flags=OUT_AFFINE)
            assert len(H) == 2 and all(np.array_equal(h, engine.to_numpy(a)) for h, a in zip(H, A)), (cv, n)
            (hx,) = engine.scalar_mult_host(cv, kn[:n], xn[:n], yn[:n], flags=OUT_AFFINE, x_only=True)
            assert np.array_equal(hx, engine.to_numpy(A[0]))
            G = engine.scalar_mult_host(cv, kn[:n], flags=OUT_AFFINE)                        # no base: the generator
            assert all(np.array_equal(g, engine.to_numpy(b)) for g, b in zip(G, engine.scalar_mult_base(cv, dev[0], flags=OUT_AFFINE)))
        if cv == SECP256K1:
            n = (1 << 19) + 5
            W = engine.scalar_mult_host(cv, kn[:n], xn[:n], yn[:n], flags=OUT_AFFINE | ALG_WINDOWED)      # per-lane tables: each side's context has its own workspace
            assert all(np.array_equal(w, engine.to_numpy(a)) for w, a in zip(W, engine.scalar_mult(cv, k[:n].contiguous(), bx[:n].contiguous(), by[:n].contiguous(), flags=OUT_AFFINE)))
    n = 2 * (1 << 19) + 3                                                                     # the context option reaches the helper context too
    pat = np.array([0, 0xffffffff, 0x80000000, 0x7fffffff, 1, 0xfffffffe], dtype=np.uint64)
    w = pat[np.random.default_rng(9).integers(0, len(pat), size=(n, 8))]
    kk = (w[:, 0::2] | (w[:, 1::2] << np.uint64(32))).astype(np.uint64)                        # carry-heavy scalars do not matter; carry-heavy COORDINATES do: take them from the ladder itself
    bx, by = engine.scalar_mult_base(P256, engine.to_device(kk), flags=OUT_AFFINE)
    engine.set_ref_square_compat(True)
    try:
        want = engine.scalar_mult(P256, k[:n].contiguous(), bx, by)
        got = engine.scalar_mult_host(P256, engine.to_numpy(k[:n]), engine.to_numpy(bx), engine.to_numpy(by))
    finally:
        engine.set_ref_square_compat(False)
    assert all(np.array_equal(g, engine.to_numpy(w_)) for g, w_ in zip(got, want))
    kn = engine.to_numpy(k[:8])
    with pytest.raises(EcsimdHipError, match="base point"):
        engine.scalar_mult_host(P256, kn, kn, kn, flags=BASE_GENERATOR)
    assert engine.lib.ecsimd_hip_scalar_mult_host(engine.ctx, C.c_int(P256), kn.ctypes.data_as(C.c_void_p), None, None, kn.ctypes.data_as(C.c_void_p), None, None, C.c_size_t(8), C.c_int(0)) == -1
    assert all(h.shape[0] == 0 for h in engine.scalar_mult_host(P256, kn[:0], kn[:0], kn[:0]))


def test_reference_register_layout_on_the_device(engine):
    """ecsimd_hip_wide4_to_lanes / _lanes_to_wide4 (r5): the reference's wide_bignum -- four lanes, limb-major u64[limb * 4 + lane] (bignum.h:99-100) -- and
    its Jacobian point of three wides, transposed on the device against numpy: an array of wides (records of 128 bytes), x / y / z of an array of points
    (records of 384 bytes, offsets 0 / 128 / 256), ragged counts, a record with padding behind the wide; and what the entry points refuse."""
    import torch
    from ecsimd_amd import EcsimdHipError
    rng = np.random.default_rng(5)
    for wides in (1, 3, 64, 1000 + 77):
        rec = rng.integers(0, 2**64, size=(wides, 3, 4, 4), dtype=np.uint64)                    # [wide][x, y, z][limb][lane]
        dev = torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(engine.tdev)
        for comp in range(3):
            got = engine.to_numpy(engine.wide4_to_lanes(dev, 384, 128 * comp, wides))
            want = rec[:, comp].transpose(0, 2, 1).reshape(4 * wides, 4)                        # element 4w + lane, limb
            assert np.array_equal(got, want), (wides, comp)
        back = torch.zeros_like(dev)
        for comp in range(3):
            engine.lanes_to_wide4(engine.wide4_to_lanes(dev, 384, 128 * comp, wides), back, 384, 128 * comp)
        assert torch.equal(back, dev)
        one = torch.from_numpy(np.ascontiguousarray(rec[:, 1]).view(np.uint8).reshape(-1)).to(engine.tdev)      # a plain array of wides
        assert np.array_equal(engine.to_numpy(engine.wide4_to_lanes(one)), rec[:, 1].transpose(0, 2, 1).reshape(4 * wides, 4))
        pad = torch.zeros(wides * 136, dtype=torch.uint8, device=engine.tdev)                   # 8 bytes of something else behind each wide: left alone
        engine.lanes_to_wide4(engine.wide4_to_lanes(one), pad, 136, 0)
        p = pad.cpu().numpy().reshape(wides, 136)
        assert np.array_equal(p[:, :128].reshape(-1), one.cpu().numpy()) and not p[:, 128:].any()
    dev = torch.zeros(1024, dtype=torch.uint8, device=engine.tdev)
    for rb, off in ((120, 0), (128, 8), (132, 0), (384, 260), (384, 12)):
        with pytest.raises(EcsimdHipError, match="128 bytes"):
            engine.wide4_to_lanes(dev, rb, off, 2)
    assert engine.wide4_to_lanes(dev, 128, 0, 0).shape[0] == 0


@pytest.mark.parametrize("cv", CURVES)
def test_wire_formats(engine, oracle, cv):
    """Device byte codecs: 32 big-endian bytes <-> limbs (serialization.h:12-48 over a batch), wide_mask_bit
    (utility.h:45-51) and SEC1 points, against numpy / the oracle's compute_y."""
    import torch
    n = 1000 + 77                                              # ragged: the last workgroup moves a partial LDS tile
    k = fill_random_np(n, SEED, 9)
    dk = engine.to_device(k)
    be = engine.to_bytes_be(dk).cpu().numpy()
    exp = np.stack([np.frombuffer(int(v).to_bytes(32, "big"), dtype=np.uint8) for v in arr_to_ints(k)])
    assert np.array_equal(be, exp)
    assert np.array_equal(engine.to_numpy(engine.from_bytes_be(torch.from_numpy(exp).to(dk.device))), k)
    for bit in (0, 1, 31, 32, 63, 64, 200, 255):
        assert engine.to_numpy(engine.mask_bit(dk, bit)).tolist() == [(v >> bit) & 1 for v in arr_to_ints(k)]
    bx, by = engine.scalar_mult_base(cv, dk, flags=2)
    xs, ys = arr_to_ints(engine.to_numpy(bx)), arr_to_ints(engine.to_numpy(by))
    p = CURVE_PARAMS[cv]["p"]
    for compressed in (False, True):
        wire = engine.sec1_encode(cv, bx, by, compressed)
        w = wire.cpu().numpy()
        for i in (0, 1, 255, 256, n - 1):
            rec = bytes(w[i])
            assert rec == ((bytes([2 | (ys[i] & 1)]) + xs[i].to_bytes(32, "big")) if compressed else (b"\x04" + xs[i].to_bytes(32, "big") + ys[i].to_bytes(32, "big")))
        dx, dy, ok = engine.sec1_decode(cv, wire, compressed)
        assert bool(ok.all()) and torch.equal(dx, bx) and torch.equal(dy, by)
        # malformed records: bad prefix, x >= p, point off the curve / non-residue
        bad = wire.clone()
        bad[3, 0] = 0x05
        bad[4, 1:33] = torch.from_numpy(np.frombuffer(p.to_bytes(32, "big"), dtype=np.uint8).copy()).to(bad.device)      # x = p
        bad[5, 32] ^= 1                                                                                                  # x + 1: off the curve or (compressed) maybe another point
        _, _, ok2 = engine.sec1_decode(cv, bad, compressed)
        ok2 = ok2.cpu().numpy()
        assert ok2[3] == 0 and ok2[4] == 0 and ok2[:3].all() and ok2[6:].all()
        if not compressed:
            assert ok2[5] == 0
        else:                                                   # compare with the oracle's square-root test for x + 1
            x5 = from_int(int.from_bytes(bytes(bad[5, 1:33].cpu().numpy()), "big"))
            _, oko = oracle.compute_y(cv, x5[None])
            assert ok2[5] == oko[0]


def test_fill_random_matches_numpy_twin(engine):
    for stream, first, clear in ((1, 0, 0), (2, 12345, 1), (3, 2**40, 8)):
        got = engine.to_numpy(engine.fill_random(1000, SEED, stream, first_index=first, clear_top_bits=clear))
        assert np.array_equal(got, fill_random_np(1000, SEED, stream, first, clear))


def test_bad_arguments_are_rejected(engine):
    import ctypes as C
    lib, ctx = engine.lib, engine.ctx
    a = engine.empty(4)
    p = C.c_void_p(a.data_ptr())
    assert lib.ecsimd_hip_mod_add(ctx, C.c_int(1 << 20), p, p, p, C.c_size_t(4)) == -1      # unknown curve / field id (small ids may be registered moduli)
    assert lib.ecsimd_hip_mod_add(ctx, C.c_int(0), C.c_void_p(0), p, p, C.c_size_t(4)) == -1  # null pointer
    assert lib.ecsimd_hip_mod_add(ctx, C.c_int(0), C.c_void_p(a.data_ptr() + 8), p, p, C.c_size_t(2)) == -1  # misaligned
    assert lib.ecsimd_hip_mod_shift_left(ctx, C.c_int(0), p, C.c_int(0), p, C.c_size_t(4)) == -1
    assert b"bad argument" in lib.ecsimd_hip_last_error(ctx)
    assert lib.ecsimd_hip_mod_add(ctx, C.c_int(0), p, p, p, C.c_size_t(0)) == 0                # empty batch is fine


# ---------------------------------------------------------------- full BASELINE sizes: properties
def _aff_ints(gpu, cv, J, idx):
    ax, ay = gpu.to_affine(cv, tuple(v[idx] for v in J))
    return [(to_int(x), to_int(y)) for x, y in zip(ax, ay)]


def test_config2_point_add_double_2pow20(engine, oracle):
    """BASELINE configs[1]: P-256 point add+double, batch 2^20 random points: TRPLU (DBLU+ZADDU) then ZDAU.
    Level J vs the oracle on a 4096-element strided sample; co-Z invariants on the whole batch."""
    import torch
    n = 1 << 20; cv = P256
    s = engine.fill_random(n, SEED, 2)
    bx, by = engine.scalar_mult_base(cv, s, flags=2)
    P = engine.from_affine(cv, bx, by)
    P0 = tuple(t.clone() for t in P)
    T = engine.trplu(cv, P)                 # T = 3P, P rewritten co-Z
    assert torch.equal(T[2], P[2])
    Z = engine.zdau(cv, T, P)               # Z = 2T + P = 7P, P rewritten co-Z
    assert torch.equal(Z[2], P[2])
    idx = np.arange(0, n, n // 4096)
    tidx = torch.from_numpy(idx).to(bx.device)
    sub = lambda pts: tuple(engine.to_numpy(t[tidx]) for t in pts)
    Rt, Put = oracle.trplu(cv, sub(P0))
    Rz, Qu = oracle.zdau(cv, Rt, Put)
    assert all(np.array_equal(u, v) for u, v in zip(sub(T), Rt)) and all(np.array_equal(u, v) for u, v in zip(sub(Z) + sub(P), Rz + Qu))
    # size-independent property: 7P computed by the formulas equals 7*P from the ladder (affine level)
    seven = engine.to_device(np.tile(from_int(7), (n, 1)))
    lx, ly = engine.scalar_mult(cv, seven, bx, by, flags=2)
    fx, fy = engine.to_affine(cv, Z)
    assert torch.equal(lx, fx) and torch.equal(ly, fy)


@pytest.mark.parametrize("cv", CURVES)
def test_x_coordinate_only_outputs(engine, oracle, cv):
    """an affine result without y (ECDH's shared secret, ECDSA's r).  Every algorithm's x-only
    output equals the x of its full output; a Jacobian result still needs y.  The x-only ladder (P-256: the ladder without
    Z, k_scalar_mult_x) is pinned to the ORACLE -- to_affine(scalar_mult(k, P)).x, curve_group.h:189-218 +
    jacobian_curve_point.h:33-42 -- on 4 096 non-degenerate lanes per entry point, and its two fixed-up scalars
    k = +-2^256 (mod n) to the big-int model."""
    import ctypes as C
    import torch
    from ecsimd_amd import OUT_AFFINE, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG
    n = (1 << 17) + 77                       # ragged; BIG's table is built from 2^16 up
    k = engine.fill_random(n, SEED, 1); s = engine.fill_random(n, SEED, 2)
    bx, by = engine.scalar_mult_base(cv, s, flags=OUT_AFFINE)
    for alg in (0, ALG_WINDOWED):
        fx, fy = engine.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE | alg)
        ox, none = engine.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE | alg, x_only=True)
        assert none is None and torch.equal(ox, fx), f"variable base, alg {alg}"
    for alg in (0, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG):
        fx, fy = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | alg)
        ox, none = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | alg, x_only=True)
        assert none is None and torch.equal(ox, fx), f"fixed base, alg {alg}"
    J = engine.scalar_mult_base(cv, k)
    ax, ay = engine.to_affine(cv, J)
    ox, none = engine.to_affine(cv, J, x_only=True)
    assert none is None and torch.equal(ox, ax) and torch.equal(fx, ax)
    # in place over the Jacobian X (the per-element kernel): still x only
    Jc = tuple(t.clone() for t in J)
    engine._call("to_affine", C.c_int(cv), *[engine._ptr(t) for t in Jc], engine._ptr(Jc[0]), C.c_void_p(0), C.c_size_t(n))
    assert torch.equal(Jc[0], ax) and torch.equal(Jc[1], J[1])
    with pytest.raises(RuntimeError):        # Jacobian output without y
        engine._call("scalar_mult_base", C.c_int(cv), engine._ptr(k), engine._ptr(ox), C.c_void_p(0), engine._ptr(Jc[2]), C.c_size_t(n), C.c_int(0))
    # every 256-bit scalar, the ladder's own degenerate ones included: on P-256 the x-only ladder runs on the odd one of
    # k mod n and n - k mod n (point.cuh scalar_mult_ladder_x) and is right where the reference's ladder is not; the
    # windowed kernels, which reduce k mod n, are the witness.  k = 0 mod n gives x = 0.
    order = CURVE_PARAMS[cv]["n"]
    edge = [0, 1, 2, 3, 4, 5, order - 2, order - 1, order, order + 1, order + 2, 2**256 - 1, 2**256 - 2, 2**255, 2**255 + 1,
            2**256 - order, 2**256 - order - 1, 2**256 - order + 1, 2 * order - 2**256, 2**64, int("55" * 32, 16), int("aa" * 32, 16)]
    m = 4096 + 32                                            # 4 106 non-degenerate lanes per form reach the oracle (at least 4 096)
    ke = fill_random_np(m, SEED, 9); ke[:len(edge)] = ints_to_arr(edge)
    kd = engine.to_device(ke)
    qx, qy = bx[:m].contiguous(), by[:m].contiguous()
    wx, _ = engine.scalar_mult(cv, kd, qx, qy, flags=OUT_AFFINE | ALG_WINDOWED)
    lx, _ = engine.scalar_mult(cv, kd, qx, qy, flags=OUT_AFFINE, x_only=True)
    if cv == P256:
        assert torch.equal(lx, wx)
        assert not lx[0].any() and not lx[8].any()          # k = 0, k = n
        for j in (77, 1, 7):                                 # one scalar for all lanes (odd, 1, n - 1)
            k1 = engine.to_numpy(kd[j:j + 1])[0]
            fx1, _ = engine.scalar_mult(cv, kd[j:j + 1].expand(m, 4).contiguous(), qx, qy, flags=OUT_AFFINE | ALG_WINDOWED)
            sx, none = engine.scalar_mult_1s(cv, k1, qx, qy, flags=OUT_AFFINE, x_only=True)
            assert none is None and torch.equal(sx, fx1)
        gx1, _ = engine.scalar_mult_base(cv, kd, flags=OUT_AFFINE | ALG_WINDOWED)
        gx2, _ = engine.scalar_mult_base(cv, kd, flags=OUT_AFFINE, x_only=True)
        assert torch.equal(gx1, gx2)
    else:                                                    # secp256k1 (a = 0) keeps the reference's ladder: equal off the degenerate scalars
        assert torch.equal(lx[len(edge):], wx[len(edge):])
    # ---- the external witness: the oracle's ladder + to_affine, x coordinate, on the non-degenerate lanes of every form
    lanes = np.arange(len(edge), m)                           # random scalars: none of the reference ladder's degenerate ones
    kn, qxn, qyn = ke[lanes], engine.to_numpy(qx)[lanes], engine.to_numpy(qy)[lanes]
    ox_var = oracle.to_affine(cv, oracle.scalar_mult(cv, kn, qxn, qyn, threads=THREADS))[0]
    assert np.array_equal(engine.to_numpy(lx)[lanes], ox_var), "scalar_mult(x_only) vs the oracle"
    c = oracle.constants(cv)
    gxn, gyn = np.tile(c["gx"], (len(lanes), 1)), np.tile(c["gy"], (len(lanes), 1))
    ox_base = oracle.to_affine(cv, oracle.scalar_mult(cv, kn, gxn, gyn, threads=THREADS))[0]
    bx_only, none = engine.scalar_mult_base(cv, kd, flags=OUT_AFFINE, x_only=True)
    assert none is None and np.array_equal(engine.to_numpy(bx_only)[lanes], ox_base), "scalar_mult_base(x_only) vs the oracle"
    k1 = ke[len(edge) + 5]                                    # one (random, non-degenerate) scalar for all lanes
    ox_1s = oracle.to_affine(cv, oracle.scalar_mult(cv, np.tile(k1, (len(lanes), 1)), qxn, qyn, threads=THREADS))[0]
    sx, none = engine.scalar_mult_1s(cv, k1, qx, qy, flags=OUT_AFFINE, x_only=True)
    assert none is None and np.array_equal(engine.to_numpy(sx)[lanes], ox_1s), "scalar_mult_1s(x_only) vs the oracle"
    # ---- k = +-2^256 (mod n): the two scalars the Z-less ladder hands to k_x_fixup (k_ladder.inc) -- and their 256-bit
    # representatives k + n where those exist -- against the independent big-int model, per-element and generator forms
    cc = (1 << 256) % order
    fix = [cc, order - cc] + [v for v in (cc + order, 2 * order - cc) if v < (1 << 256)]
    kf = engine.to_device(ints_to_arr(fix))
    fqx, fqy = qx[:len(fix)].contiguous(), qy[:len(fix)].contiguous()
    fx_var, _ = engine.scalar_mult(cv, kf, fqx, fqy, flags=OUT_AFFINE, x_only=True)
    fx_base, _ = engine.scalar_mult_base(cv, kf, flags=OUT_AFFINE, x_only=True)
    pts = list(zip(arr_to_ints(engine.to_numpy(fqx)), arr_to_ints(engine.to_numpy(fqy))))
    G = (CURVE_PARAMS[cv]["gx"], CURVE_PARAMS[cv]["gy"])
    if cv == P256:                                           # secp256k1 keeps the reference ladder, whose own degenerate scalars these are
        assert arr_to_ints(engine.to_numpy(fx_var)) == [ec_mul(cv, kk, P)[0] for kk, P in zip(fix, pts)]
        assert arr_to_ints(engine.to_numpy(fx_base)) == [ec_mul(cv, kk, G)[0] for kk in fix]
        for kk in fix[:2]:
            s1, _ = engine.scalar_mult_1s(cv, ints_to_arr([kk])[0], fqx, fqy, flags=OUT_AFFINE, x_only=True)
            assert arr_to_ints(engine.to_numpy(s1)) == [ec_mul(cv, kk, P)[0] for P in pts]


@pytest.mark.parametrize("cv,log2n", [(P256, 22), (SECP256K1, 22)])
def test_config3_and_5_fixed_base_2pow22(engine, oracle, cv, log2n):
    """BASELINE configs[2] and [4]: k*G for 2^22 random scalars.  Oracle on a strided sample (level J)
    plus linearity on the whole batch: (k + 1)*G == k*G + G, checked through ADD_Z2_1 at affine level."""
    import torch
    n = 1 << log2n
    k = engine.fill_random(n, SEED, 1)
    J = engine.scalar_mult_base(cv, k)
    idx = np.arange(0, n, n // 2048)
    tidx = torch.from_numpy(idx).to(k.device)
    kn = engine.to_numpy(k[tidx]); c = CURVE_PARAMS[cv]
    exp = oracle.scalar_mult(cv, kn, ints_to_arr([c["gx"]] * len(idx)), ints_to_arr([c["gy"]] * len(idx)), threads=THREADS)
    assert all(np.array_equal(engine.to_numpy(t[tidx]), v) for t, v in zip(J, exp))
    # linearity: k*G + G == (k+1)*G  (k+1 computed on the host side of the device: add with carry)
    one = engine.to_device(np.tile(from_int(1), (n, 1)))
    k1, _ = engine.add(k, one)
    J1 = engine.scalar_mult_base(cv, k1)
    Gm = engine.from_affine(cv, engine.to_device(ints_to_arr([c["gx"]] * n)), engine.to_device(ints_to_arr([c["gy"]] * n)))
    S = engine.add_z2_1(cv, J, (Gm[0], Gm[1]))
    ax, ay = engine.to_affine(cv, S); bx, by = engine.to_affine(cv, J1)
    assert torch.equal(ax, bx) and torch.equal(ay, by)
    # BASELINE configs[2] names the 4-bit window table in LDS: that kernel and its two siblings (signed 7-bit windows in LDS,
    # 20-bit windows over the 436 MB table in device memory) at the SAME full size, every lane against the ladder's affine
    # result -- all 13 x 2^19 entries of the big table and all 64 x 16 / 37 x 64 of the LDS ones are in play at 2^22 scalars --
    # and a strided sample against the oracle.
    lx, ly = engine.to_affine(cv, J)
    ex, ey = oracle.to_affine(cv, exp)
    for alg in (ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG):
        wx, wy = engine.scalar_mult_base(cv, k, flags=OUT_AFFINE | alg)
        assert torch.equal(wx, lx) and torch.equal(wy, ly), ("windowed fixed-base kernel differs from the ladder", alg)
        assert np.array_equal(engine.to_numpy(wx[tidx]), ex) and np.array_equal(engine.to_numpy(wy[tidx]), ey)


def test_config4_variable_base_shard_2pow21(engine, oracle):
    """BASELINE configs[3]: variable base, 2^24 over 8 GPUs = 2^21 per GPU.  This is rank 3's shard:
    inputs regenerated from (seed, global index), oracle on a strided sample, and the round trip
    k^-1 * (k * P) == P on the whole shard (scalars made invertible mod n by construction)."""
    import torch
    cv = P256; n = 1 << 21; rank = 3; order = CURVE_PARAMS[cv]["n"]
    k = engine.fill_random(n, SEED, 1, first_index=rank * n)
    s = engine.fill_random(n, SEED, 2, first_index=rank * n)
    bx, by = engine.scalar_mult_base(cv, s, flags=2)
    J = engine.scalar_mult(cv, k, bx, by)
    idx = np.arange(0, n, n // 1024); tidx = torch.from_numpy(idx).to(k.device)
    kn, bxn, byn = (engine.to_numpy(t[tidx]) for t in (k, bx, by))
    assert np.array_equal(kn, fill_random_np(n, SEED, 1, first_index=rank * n)[idx])
    exp = oracle.scalar_mult(cv, kn, bxn, byn, threads=THREADS)
    assert all(np.array_equal(engine.to_numpy(t[tidx]), v) for t, v in zip(J, exp))
    # round trip on a 2^16 slice (host computes the modular inverses of the scalars)
    m = 1 << 16
    ks = arr_to_ints(engine.to_numpy(k[:m]))
    kinv = ints_to_arr([pow(v % order, -1, order) for v in ks])
    ax, ay = engine.to_affine(cv, tuple(t[:m].contiguous() for t in J))
    rx, ry = engine.scalar_mult(cv, engine.to_device(kinv), ax, ay, flags=2)
    assert torch.equal(rx, bx[:m]) and torch.equal(ry, by[:m])


# ---------------------------------------------------------------- literal parity with the compiled reference
def _carry_heavy_field_elements(cv, n, seed):
    p = CURVE_PARAMS[cv]["p"]
    return ints_to_arr([v % p for v in arr_to_ints(structured_words(n, seed=seed))])


@pytest.mark.parametrize("cv", CURVES)
def test_ref_square_compat_is_the_reference_bug_for_bug(engine, gpu, oracle, oracle_faithful, golden, cv):
    """ECSIMD_HIP_REF_SQUARE_COMPAT / ecsimd_hip_set_ref_square_compat: square with mul.h:160-212 AS WRITTEN (the carry
    dropped at mul.h:186-190 included).  Pinned by the vectors minted from the compiled reference
    (ref_vectors.json "square_defect": reference_square, p256_reference_mgry_sqr) and, on carry-heavy operands where
    the defect is frequent, by the bug-for-bug oracle (itself proven equal to the compiled reference,
    tests/test_oracle.py::test_structured_operands_vs_live_reference) through every layer above the squaring."""
    d = golden["square_defect"]; x = hexes_to_arr(d["a"])
    a = structured_words(100000)
    ar = _carry_heavy_field_elements(cv, 20000, 3); br = np.roll(ar, 5, axis=0)
    m = 2048
    k = fill_random_np(m, SEED, 7); k[:8, 0] &= np.uint64(~np.uint64(1))            # some even scalars: the ADD_Z2_1 tail
    engine.set_ref_square_compat(True)
    try:
        assert arr_to_hexes(gpu.square(x), 8) == d["reference_square"]
        if cv == P256:
            assert arr_to_hexes(gpu.mgry_sqr(P256, x)) == d["p256_reference_mgry_sqr"]
        sq = gpu.square(a)
        assert np.array_equal(sq, oracle_faithful.square(a))
        wrong = (sq != oracle.square(a)).any(axis=1)
        assert 1000 < wrong.sum() < len(a)                                             # the defect is really exercised
        got = gpu.mgry_sqr(cv, ar)
        assert np.array_equal(got, oracle_faithful.mgry_sqr(cv, ar)) and (got != oracle.mgry_sqr(cv, ar)).any()
        assert np.array_equal(gpu.mgry_mul(cv, ar, br), oracle.mgry_mul(cv, ar, br))   # mul() has no defect
        e = from_int(0x1234567890abcdef_0fedcba987654321_00000000ffffffff_8000000000000001)
        assert np.array_equal(gpu.mgry_pow(cv, ar[:m], e), oracle_faithful.mgry_pow(cv, ar[:m], e))
        assert np.array_equal(gpu.gfp_inverse(cv, ar[:m]), oracle_faithful.gfp_inverse(cv, ar[:m]))
        s_, ok = gpu.gfp_sqrt(cv, ar[:m]); so, oko = oracle_faithful.gfp_sqrt(cv, ar[:m])
        assert np.array_equal(s_, so) and np.array_equal(ok, oko)
        # the formulas are plain field arithmetic: carry-heavy (x, y) pairs need not be curve points to compare DAGs
        P = oracle.from_affine(cv, ar[:m], br[:m])
        (Rg, Pg), (Ro, Po) = gpu.trplu(cv, P), oracle_faithful.trplu(cv, P)
        assert all(np.array_equal(u, v) for u, v in zip(Rg + Pg, Ro + Po))
        assert any((u != v).any() for u, v in zip(Ro + Po, sum(oracle.trplu(cv, P), ())))        # ... and differ from the exact DAG (DBLU squares x and y)
        (Zg, Qg), (Zo, Qo) = gpu.zdau(cv, Ro, Po), oracle_faithful.zdau(cv, Ro, Po)
        assert all(np.array_equal(u, v) for u, v in zip(Zg + Qg, Zo + Qo))
        assert all(np.array_equal(u, v) for u, v in zip(gpu.add_z2_1(cv, Zo, (P[0], P[1])), oracle_faithful.add_z2_1(cv, Zo, (P[0], P[1]))))
        Jf = oracle_faithful.scalar_mult(cv, k, ar[:m], br[:m], threads=THREADS)
        Jg = gpu.scalar_mult(cv, k, ar[:m], br[:m])
        assert all(np.array_equal(u, v) for u, v in zip(Jg, Jf))
        assert any((u != v).any() for u, v in zip(Jf, oracle.scalar_mult(cv, k, ar[:m], br[:m], threads=THREADS)))
        af = oracle_faithful.to_affine(cv, Jf)
        assert all(np.array_equal(u, v) for u, v in zip(gpu.to_affine(cv, Jf), af))
        assert all(np.array_equal(u, v) for u, v in zip(gpu.scalar_mult(cv, k, ar[:m], br[:m], affine=True), af))
        with pytest.raises(EcsimdHipError):                      # the windowed algorithms are not the reference's DAG
            engine.scalar_mult_base(cv, engine.to_device(k), flags=OUT_AFFINE | ALG_WINDOWED)
    finally:
        engine.set_ref_square_compat(False)
    # the per-call flag does the same for the ladder, and the default stays exact
    dk, dx, dy = (engine.to_device(v) for v in (k, ar[:m], br[:m]))
    Jflag = engine.scalar_mult(cv, dk, dx, dy, flags=REF_SQUARE_COMPAT)
    assert all(np.array_equal(engine.to_numpy(u), v) for u, v in zip(Jflag, Jf))
    Jexact = engine.scalar_mult(cv, dk, dx, dy)
    assert all(np.array_equal(engine.to_numpy(u), v) for u, v in zip(Jexact, oracle.scalar_mult(cv, k, ar[:m], br[:m], threads=THREADS)))
    assert arr_to_hexes(gpu.square(x), 8) == d["exact_square"]


@pytest.mark.parametrize("cv", CURVES)
def test_ladder_vs_the_live_reference_2pow20(engine, oracle, reference, openssl, cv):
    """2^20 random (scalar, point) pairs through the COMPILED reference (oracle/_ref, skipped where it was not built) and
    through the HIP ladder.  With ECSIMD_HIP_REF_SQUARE_COMPAT not one lane may differ.  Without it the HIP result is the exact
    k*P: every lane that differs from the reference must be confirmed by libcrypto -- an implementation that shares no
    code or algorithm with either -- and by the exact oracle."""
    import torch
    n = 1 << 20
    k = engine.fill_random(n, SEED, 1, first_index=5 * n); s = engine.fill_random(n, SEED, 2, first_index=5 * n)
    bx, by = engine.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED_BIG)
    kn, bxn, byn = (engine.to_numpy(t) for t in (k, bx, by))
    ref = reference.scalar_mult(cv, kn, bxn, byn, threads=THREADS)
    compat = engine.scalar_mult(cv, k, bx, by, flags=REF_SQUARE_COMPAT)
    for name, g, r in zip("XYZ", compat, ref):
        assert np.array_equal(engine.to_numpy(g), r), f"compat ladder {name} differs from the compiled reference"
    J = engine.scalar_mult(cv, k, bx, by)
    got = [engine.to_numpy(t) for t in J]
    bad = np.nonzero((got[0] != ref[0]).any(axis=1) | (got[1] != ref[1]).any(axis=1) | (got[2] != ref[2]).any(axis=1))[0]
    assert len(bad) < 64                                           # ~3e-6 of the lanes
    if len(bad):
        ax, ay = engine.to_affine(cv, tuple(t[torch.from_numpy(bad).to(t.device)].contiguous() for t in J))
        vx, vy, inf = openssl.scalar_mult(cv, kn[bad], bxn[bad], byn[bad], threads=1)
        assert not inf.any() and np.array_equal(engine.to_numpy(ax), vx) and np.array_equal(engine.to_numpy(ay), vy), "libcrypto sides with the reference"
        ex = oracle.scalar_mult(cv, kn[bad], bxn[bad], byn[bad], threads=1)
        assert all(np.array_equal(g[bad], e) for g, e in zip(got, ex))
        rx, ry = reference.to_affine(cv, tuple(v[bad] for v in ref))             # the reference's own point on those lanes is NOT k*P
        assert ((rx != vx) | (ry != vy)).any(axis=1).all()
    print(f"curve {cv}: {len(bad)} of {n} lanes differ from the compiled reference without the flag, 0 with it")


@pytest.mark.parametrize("cv", CURVES)
def test_field_ops_on_every_digit_pattern(gpu, oracle, cv):
    """The default (exact) field arithmetic on the same exhaustive family -- every operand with digits from {0, 1, 7fffffff, 80000000,
    fffffffe, ffffffff}, brought below p like the reference's operands are (sub.h:46-69) -- against the oracle: 256 x 256 -> 512
    multiply and square, Montgomery multiply and square, modular add / subtract / double.  1 679 616 operand pairs per operation."""
    from test_oracle import digit_pattern_operands
    a = digit_pattern_operands()
    b = np.roll(a, 271828, axis=0)
    assert np.array_equal(gpu.mul(a, b), oracle.mul(a, b))
    p = np.tile(oracle.constants(cv)["p"], (len(a), 1))
    ar, br = oracle.sub_if_above(a, p), oracle.sub_if_above(b, p)
    assert np.array_equal(gpu.sub_if_above(a, p), ar)
    for name in ("mgry_mul", "mod_add", "mod_sub"):
        assert np.array_equal(getattr(gpu, name)(cv, ar, br), getattr(oracle, name)(cv, ar, br)), name
    assert np.array_equal(gpu.mgry_sqr(cv, ar), oracle.mgry_sqr(cv, ar))
    assert np.array_equal(gpu.mod_shift_left(cv, ar, 1), oracle.mod_shift_left(cv, ar, 1))
    assert np.array_equal(gpu.mod_shift_left(cv, ar, 2), oracle.mod_shift_left(cv, ar, 2))


def test_ref_square_compat_every_digit_pattern(engine, gpu, oracle, oracle_faithful):
    """The hand-laid reference squaring (field.cuh sqr8_ref: carry of the doubled product = carry-out of the multiply, the dropped
    carry = a 64-bit add without carry-out) on EVERY operand whose eight 32-bit digits are drawn from {0, 1, 7fffffff, 80000000,
    fffffffe, ffffffff}: 6^8 = 1 679 616 operands, the ones that drive its rare paths (a 33-bit addend, both row-end carries,
    the unnormalised digit wrapping the last diagonal).  Bit for bit against the bug-for-bug restatement of mul.h:160-212,
    which tests/test_oracle.py pins to the compiled reference; and the exact squaring on the same operands against mul(a, a)."""
    pat = np.array([0, 1, 0x7fffffff, 0x80000000, 0xfffffffe, 0xffffffff], dtype=np.uint64)
    idx = np.indices((6,) * 8).reshape(8, -1).T                              # every digit combination
    w = pat[idx]
    a = (w[:, 0::2] | (w[:, 1::2] << np.uint64(32))).astype(np.uint64)
    assert a.shape == (6 ** 8, 4)
    exact = gpu.square(a)
    assert np.array_equal(exact, oracle.square(a))
    engine.set_ref_square_compat(True)
    try:
        got = gpu.square(a)
    finally:
        engine.set_ref_square_compat(False)
    assert np.array_equal(got, oracle_faithful.square(a))
    wrong = int((got != exact).any(axis=1).sum())
    assert wrong > 100000, wrong                                             # the defect is everywhere in this family
    print(f"{len(a)} digit-pattern operands: the reference's square() differs from a^2 on {wrong} of them; the compat kernel reproduces every one")


def test_operands_must_agree_on_the_batch_length(engine):
    """The C ABI takes one length for all operands; the binding refuses tensors that disagree (they would be read or
    written out of bounds on the device)."""
    k = engine.fill_random(64, SEED, 1); x = engine.fill_random(32, SEED, 2, clear_top_bits=1)
    with pytest.raises(EcsimdHipError):
        engine.scalar_mult(P256, k, x, x)
    with pytest.raises(EcsimdHipError):
        engine.mgry_mul(P256, k, x)
    with pytest.raises(EcsimdHipError):
        engine.scalar_mult(P256, x, x, x, out=[engine.empty(16) for _ in range(3)])
    assert engine.mgry_mul(P256, x, x).shape == (32, 4)         # and the refused calls left no state behind


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_device_group_behind_the_c_abi(engine, oracle, devices):
    """ecsimd_hip_group_*: contiguous shards, one context per member, results gathered to member 0.  A one-GPU box
    can only build groups on device 0: [0] is the 1-device group of SURVEY.md 8(e); listing the device two or three
    times exercises the multi-member bookkeeping (uneven shards, staging, the gather into member 0's arrays) with
    device copies where RCCL -- which refuses two ranks on one GPU -- would run on a real node."""
    _device_group_checks(engine, oracle, devices, expect_rccl=False)


FAKE_RCCL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfake_rccl.so")


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_device_group_rccl_branch_on_a_double(engine, oracle, devices, monkeypatch):
    """The RCCL branch of ecsimd_hip_group_scalar_mult (group.hip: ncclCommInitAll, the grouped ncclRecv / ncclSend exchange) had never
    executed: every box has one GPU and the real RCCL refuses two ranks on one device.  Here the library is pointed at the test double
    tests/fake_rccl (ECSIMD_HIP_RCCL_LIB) and told to take the RCCL branch although the members share device 0
    (ECSIMD_HIP_GROUP_FORCE_RCCL=1): the same checks as above -- uneven and empty shards, Jacobian / affine / x-only, the host-array form,
    back-to-back calls without a sync, NO_GATHER -- must pass with uses_rccl == True, and the double's counters must show that every
    shard went through a paired send / receive.  The double is strict (an unmatched or mismatched pair, a wrong current device, a call
    outside a group is an error) and models stream order; it says nothing about RCCL's own transport."""
    import ctypes
    if not os.path.exists(FAKE_RCCL):
        pytest.fail("tests/fake_rccl/libfake_rccl.so is missing: __graft_entry__.build() makes it")
    monkeypatch.setenv("ECSIMD_HIP_RCCL_LIB", FAKE_RCCL)
    monkeypatch.setenv("ECSIMD_HIP_GROUP_FORCE_RCCL", "1")
    fake = ctypes.CDLL(FAKE_RCCL)
    stats = lambda: (lambda a: (fake.fake_rccl_stats(a), list(a))[1])((ctypes.c_ulonglong * 5)())
    before = stats()
    _device_group_checks(engine, oracle, devices, expect_rccl=True)
    after = stats()
    sends, recvs, groups, nbytes, comms = (a - b for a, b in zip(after, before))
    G = len(devices); n = 10007
    assert comms == G and groups == 8                                   # one communicator per member; the eight gathering calls of the checks
    # per gathering call one pair per (sending member, output array): 3 Jacobian, 2 affine, 1 x-only; the 2-element batch has one sender
    pairs = (G - 1) * (3 + 2 + 3 + 1 + 1 + 3 + 3) + 3
    assert sends == recvs == pairs
    from ecsimd_amd import shard_range_c
    per = lambda total: sum(shard_range_c(total, m, G)[1] for m in range(1, G))          # elements that travel: every shard but member 0's
    assert nbytes == 32 * (per(n) * (3 + 2 + 3 + 1 + 1 + 3 + 3) + per(2) * 3)


def _device_group_checks(engine, oracle, devices, expect_rccl):
    import torch
    from ecsimd_amd import DeviceGroup, shard_range_c
    cv = P256; n = 10007; G = len(devices)
    grp = DeviceGroup(devices)
    try:
        assert grp.size == G and grp.uses_rccl is expect_rccl
        k = engine.fill_random(n, SEED, 1, first_index=77); s_ = engine.fill_random(n, SEED, 2, first_index=77)
        bx, by = engine.scalar_mult_base(cv, s_, flags=OUT_AFFINE | ALG_WINDOWED_BIG)
        exp = engine.scalar_mult(cv, k, bx, by)
        spans = [shard_range_c(n, m, G) for m in range(G)]
        cut = lambda t: [t[f:f + c].contiguous() for f, c in spans]
        got, gather_ms = grp.scalar_mult(cv, cut(k), cut(bx), cut(by), n)
        assert all(torch.equal(a, b) for a, b in zip(got, exp)) and (gather_ms >= 0 if G > 1 else gather_ms == -1)     # one member: nothing to gather
        (ax, ay), _ = grp.scalar_mult(cv, cut(k), cut(bx), cut(by), n, flags=OUT_AFFINE | ALG_WINDOWED)
        ex, ey = engine.to_affine(cv, exp)
        assert torch.equal(ax, ex) and torch.equal(ay, ey)
        # host-array form, checked against the oracle on a sample as well
        kn, xn, yn = (engine.to_numpy(t) for t in (k, bx, by))
        H = grp.scalar_mult_host(cv, kn, xn, yn)
        assert all(np.array_equal(h, engine.to_numpy(e)) for h, e in zip(H, exp))
        idx = np.arange(0, n, 97)
        assert all(np.array_equal(h[idx], o) for h, o in zip(H, oracle.scalar_mult(cv, kn[idx], xn[idx], yn[idx], threads=THREADS)))
        # fewer elements than members: empty shards
        small = grp.scalar_mult_host(cv, kn[:2], xn[:2], yn[:2])
        assert all(np.array_equal(h, engine.to_numpy(e)[:2]) for h, e in zip(small, exp))
        with pytest.raises(EcsimdHipError):
            grp.scalar_mult(cv, cut(k), cut(bx), [None] * G, n)
        # x coordinate only through the group (oy = NULL with OUT_AFFINE: P-256 runs the ladder without Z on every member)
        (x1,), _ = grp.scalar_mult(cv, cut(k), cut(bx), cut(by), n, flags=OUT_AFFINE, x_only=True)
        assert torch.equal(x1, ex)
        (hx,) = grp.scalar_mult_host(cv, kn, xn, yn, flags=OUT_AFFINE, x_only=True)
        assert np.array_equal(hx, engine.to_numpy(ex))
        with pytest.raises(EcsimdHipError):
            grp.enqueue(cv, cut(k), cut(bx), cut(by), grp.alloc_outputs(n)[:1], n)      # a Jacobian result needs all three arrays
        # calls may follow one another without a sync: each member's next ladder waits for the exchange that reads its
        # staging.  Two different batches back to back into two sets of arrays, one sync at the end.
        k2 = engine.fill_random(n, SEED, 5, first_index=11)
        exp2 = engine.scalar_mult(cv, k2, bx, by)
        torch.cuda.synchronize()
        o1, o2 = grp.alloc_outputs(n), grp.alloc_outputs(n)
        grp.enqueue(cv, cut(k), cut(bx), cut(by), o1, n)
        grp.enqueue(cv, cut(k2), cut(bx), cut(by), o2, n)
        assert grp.sync() >= (0 if G > 1 else -1)
        assert all(torch.equal(a, b) for a, b in zip(o1, exp)) and all(torch.equal(a, b) for a, b in zip(o2, exp2))
        # the ladders without the exchange: member 0's slice is in place, the rest of the arrays is left alone
        o3 = [torch.full_like(t, 7) for t in o1]
        grp.enqueue(cv, cut(k), cut(bx), cut(by), o3, n, flags=GROUP_NO_GATHER)
        grp.sync()
        f0, c0 = spans[0]
        assert all(torch.equal(a[f0:f0 + c0], b[f0:f0 + c0]) for a, b in zip(o3, exp))
        assert G == 1 or all(bool((a[f0 + c0:] == 7).all()) for a in o3)
        assert all(grp.member_ms(m) > 0 for m in range(G)) and grp.rccl_version == (99999 if expect_rccl else 0)
        assert torch.cuda.current_device() == 0
        if G == 1:
            # what one GPU can run of the RCCL side: dlopen, a one-rank communicator, the gather's own send / recv calls
            grp.rccl_selftest(1 << 16)
    finally:
        grp.close()


@pytest.mark.parametrize("cv", CURVES)
def test_invalid_public_keys_are_rejected(engine, cv):
    """double_scalar_mult / ecdsa_verify_rx validate Q: the point at infinity (0, 0), a point off the curve and a
    coordinate >= p give (0, 0), finite = 0, ok = 0 -- and leave their neighbours alone.  A fresh context below 2^16
    elements takes the LDS-table route for u1*G (no 436 MB table is built for a small batch): same results."""
    import torch
    from ecsimd_amd import Engine
    c = CURVE_PARAMS[cv]; p = c["p"]; n = 300
    u1 = engine.fill_random(n, SEED, 71); u2 = engine.fill_random(n, SEED, 72)
    qx, qy = engine.scalar_mult_base(cv, engine.fill_random(n, SEED, 73), flags=OUT_AFFINE | ALG_WINDOWED)
    gx, gy, gfin = engine.double_scalar_mult(cv, u1, u2, qx, qy)
    assert bool(gfin.all()) and bool(engine.on_curve(cv, qx, qy).all())
    bx, by = engine.to_numpy(qx).copy(), engine.to_numpy(qy).copy()
    bx[3] = 0; by[3] = 0                                             # infinity
    by[10] = from_int((to_int(by[10]) + 1) % p)                      # off the curve
    bx[20] = from_int(p + 1)                                         # x >= p (x = 1 mod p may well be on the curve: the range check must catch it)
    by[30] = from_int(p - to_int(by[30]))                            # -Q: a VALID key, another point
    bad = [3, 10, 20]
    dqx, dqy = engine.to_device(bx), engine.to_device(by)
    okc = engine.to_numpy(engine.on_curve(cv, dqx, dqy))
    assert [i for i in range(n) if not okc[i]] == bad
    rx, ry, fin = engine.double_scalar_mult(cv, u1, u2, dqx, dqy)
    rxn, ryn, finn = engine.to_numpy(rx), engine.to_numpy(ry), engine.to_numpy(fin)
    same = np.ones(n, dtype=bool); same[bad + [30]] = False
    assert np.array_equal(rxn[same], engine.to_numpy(gx)[same]) and np.array_equal(ryn[same], engine.to_numpy(gy)[same]) and finn[same].all()
    assert not finn[bad].any() and not rxn[bad].any() and not ryn[bad].any() and finn[30] == 1
    r = engine.to_device(ints_to_arr([to_int(v) % c["n"] for v in engine.to_numpy(gx)]))            # r = x mod n of the honest sums
    ok = engine.to_numpy(engine.ecdsa_verify_rx(cv, u1, u2, dqx, dqy, r))
    assert ok[same].all() and not ok[bad].any() and ok[30] == 0
    # x-only output, no finite array
    rx2, _, _ = engine.double_scalar_mult(cv, u1, u2, dqx, dqy, x_only=True)
    assert torch.equal(rx2, rx)
    fresh = Engine(0)
    try:
        fx, fy, ffin = fresh.double_scalar_mult(cv, u1, u2, dqx, dqy)
        torch.cuda.synchronize()
        assert torch.equal(fx, rx) and torch.equal(fy, ry) and torch.equal(ffin, fin)
        assert torch.equal(fresh.ecdsa_verify_rx(cv, u1, u2, dqx, dqy, r), engine.ecdsa_verify_rx(cv, u1, u2, dqx, dqy, r))
    finally:
        fresh.close()


def test_workspace_growth_is_refused_during_stream_capture():
    """Growing the context workspace frees the old block (pointers an earlier hipGraph captured would dangle) and needs a
    synchronisation a capture cannot have: a call that would have to grow it -- or build a window table -- while its
    stream is being captured returns an error; after a warm-up at that size the same call captures and replays."""
    import torch
    from ecsimd_amd import Engine
    eng = Engine(0)
    try:
        n = 1 << 12
        k = eng.fill_random(n, SEED, 81); bx, by = eng.scalar_mult_base(P256, eng.fill_random(n, SEED, 82), flags=OUT_AFFINE)
        out = [eng.empty(n), eng.empty(n), None]
        eager = eng.scalar_mult(P256, k, bx, by)                     # Jacobian ladder: no workspace
        torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        refused = []
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for flags in (OUT_AFFINE | ALG_WINDOWED, OUT_AFFINE | ALG_WINDOWED_SIGNED):
                    try:
                        eng.scalar_mult_base(P256, k, flags=flags, out=out)          # needs a table + a workspace
                    except EcsimdHipError as exc:
                        refused.append(str(exc))
                try:
                    eng.scalar_mult(P256, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED, out=out)    # 1 408 B of workspace per element
                except EcsimdHipError as exc:
                    refused.append(str(exc))
                J = eng.scalar_mult(P256, k, bx, by)                                  # capturable as it is
        torch.cuda.synchronize()
        assert len(refused) == 3 and all("capture" in m for m in refused), refused
        g.replay(); torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(J, eager))
        want = eng.scalar_mult(P256, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)     # warm-up sizes the workspace ...
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g2, stream=side):
                eng.scalar_mult(P256, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED, out=out)     # ... and now it captures
        g2.replay(); torch.cuda.synchronize()
        assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])
    finally:
        eng.close()


@pytest.mark.parametrize("cv", CURVES)
def test_inversion_by_division_steps(engine, gpu, oracle, cv):
    """GFp::inverse (gfp.h:42-44) on the device is the Bernstein-Yang division-step inversion, not a^(p-2): the inverse of a residue is
    unique, so it must equal the reference's power bit for bit.  Edge operands (1, 2, p - 1, (p +- 1)/2, powers of two, values with
    long runs of zero and one bits, carry-heavy digits), 20 000 random ones against Python's pow(x, -1, p), the oracle on a sample,
    0 -> 0, and both device paths: the batched kernel (simultaneous inversion, one division-step inversion per lane) and the
    per-element kernel the ABI uses when out aliases a."""
    import ctypes as C
    pr = CURVE_PARAMS[cv]["p"]; Rinv = pow(1 << 256, -1, pr); Rr = (1 << 256) % pr
    edge = [1, 2, 3, pr - 1, pr - 2, (pr - 1) // 2, (pr + 1) // 2, 2**255, 2**254 + 1, 2**128, 2**96 - 1, 2**224, 2**32, 2**30, 2**30 - 1, 2**60 + 1,
            (1 << 255) - 1, int("5" * 64, 16) % pr, int("a" * 64, 16) % pr, 0]
    rng = np.random.default_rng(77 + cv)
    vals = edge + [int.from_bytes(rng.bytes(32), "big") % pr for _ in range(20000)] + arr_to_ints(_carry_heavy_field_elements(cv, 2000, 21))
    a = ints_to_arr(vals)
    want = ints_to_arr([(pow(v * Rinv % pr, -1, pr) * Rr % pr) if v else 0 for v in vals])      # Montgomery form in, Montgomery form out
    got = gpu.gfp_inverse(cv, a)                                                                  # batched kernel
    assert np.array_equal(got, want)
    assert np.array_equal(got[:600], oracle.gfp_inverse(cv, a[:600]))
    t = engine.to_device(a)                                                                       # in place: the per-element kernel
    engine._call("gfp_inverse", C.c_int(cv), engine._ptr(t), engine._ptr(t), C.c_size_t(len(vals)))
    assert np.array_equal(engine.to_numpy(t), want)
    # a * a^-1 = 1 on the whole batch (Montgomery form of 1 = R mod p), zero excepted
    prod = gpu.mgry_mul(cv, a, got)
    one = ints_to_arr([Rr if v else 0 for v in vals])
    assert np.array_equal(prod, one)
