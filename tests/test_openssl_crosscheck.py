"""Independent cross-check against OpenSSL's libcrypto (oracle/ossl_check.c; SURVEY.md 8(f) rank 4, the
competitor of benchs/p256_ref.cpp:55-91).  OpenSSL shares no code and no algorithm with aguinet/ecsimd or with
the restatement, so agreement at the affine level (level A) is evidence that does not lean on the oracle.
The CPU tests pin the oracle; the GPU tests (-m gpu) pin the HIP path."""
import os

import numpy as np
import pytest

from helpers import CURVE_PARAMS, P256, SECP256K1, SEED, from_int, to_int, ints_to_arr, fill_random_np

CURVES = [P256, SECP256K1]
THREADS = min(16, os.cpu_count() or 1)


def _base(cv, n):
    c = CURVE_PARAMS[cv]
    return ints_to_arr([c["gx"]] * n), ints_to_arr([c["gy"]] * n)


@pytest.mark.parametrize("cv", CURVES)
def test_oracle_ladder_agrees_with_openssl(oracle, openssl, cv):
    n = 96
    order = CURVE_PARAMS[cv]["n"]
    k = fill_random_np(n, SEED, 41)
    k[:6] = ints_to_arr([1, 2, 3, order - 2, order + 1, 2**256 - 1])                # the ladder takes any 256-bit scalar
    s = fill_random_np(n, SEED, 42)
    gx, gy = _base(cv, n)
    bx, by, inf = openssl.scalar_mult_base(cv, s)
    assert not inf.any()
    ox, oy = oracle.to_affine(cv, oracle.scalar_mult(cv, s, gx, gy, threads=THREADS))
    assert np.array_equal(ox, bx) and np.array_equal(oy, by)                        # k*G
    vx, vy, inf = openssl.scalar_mult(cv, k, bx, by, threads=THREADS)
    assert not inf.any()
    ox, oy = oracle.to_affine(cv, oracle.scalar_mult(cv, k, bx, by, threads=THREADS))
    assert np.array_equal(ox, vx) and np.array_equal(oy, vy)                        # k*P, lane-distinct P


def test_openssl_flags_infinity_and_rejects_points_off_the_curve(openssl):
    c = CURVE_PARAMS[P256]
    gx, gy = _base(P256, 2)
    _, _, inf = openssl.scalar_mult(P256, ints_to_arr([c["n"], 5]), gx, gy)
    assert list(inf) == [1, 0]
    bad = gy.copy(); bad[0, 0] ^= np.uint64(1)
    with pytest.raises(AssertionError):
        openssl.scalar_mult(P256, ints_to_arr([5, 5]), gx, bad)


@pytest.mark.gpu
@pytest.mark.parametrize("cv", CURVES)
def test_gpu_scalar_mult_agrees_with_openssl(engine, openssl, cv):
    """Variable base (the reference ladder + batched to_affine), fixed base (both window kernels) and
    u1*G + u2*Q on the HIP path against libcrypto, 4 096 lane-distinct inputs each."""
    from ecsimd_amd import OUT_AFFINE, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG
    n = 4096
    k = fill_random_np(n, SEED, 43); s = fill_random_np(n, SEED, 44); u1 = fill_random_np(n, SEED, 45)
    order = CURVE_PARAMS[cv]["n"]
    k[:4] = ints_to_arr([1, 2, order - 2, 2**256 - 1]); s[:2] = ints_to_arr([1, order - 2])
    dk, ds, du1 = engine.to_device(k), engine.to_device(s), engine.to_device(u1)
    bx, by, inf = openssl.scalar_mult_base(cv, s, threads=THREADS)
    assert not inf.any()
    for flags in (OUT_AFFINE, OUT_AFFINE | ALG_WINDOWED, OUT_AFFINE | ALG_WINDOWED_SIGNED, OUT_AFFINE | ALG_WINDOWED_BIG):
        gx_, gy_ = engine.scalar_mult_base(cv, ds, flags=flags)
        assert np.array_equal(engine.to_numpy(gx_), bx) and np.array_equal(engine.to_numpy(gy_), by), flags
    dbx, dby = engine.to_device(bx), engine.to_device(by)
    vx, vy, inf = openssl.scalar_mult(cv, k, bx, by, threads=THREADS)
    assert not inf.any()
    for flags in (OUT_AFFINE, OUT_AFFINE | ALG_WINDOWED):                           # reference ladder; per-lane window tables
        px, py = engine.scalar_mult(cv, dk, dbx, dby, flags=flags)
        assert np.array_equal(engine.to_numpy(px), vx) and np.array_equal(engine.to_numpy(py), vy), flags
    xo, none = engine.scalar_mult(cv, dk, dbx, dby, flags=OUT_AFFINE, x_only=True)   # P-256: the ladder without Z (ECDH's x)
    assert none is None and np.array_equal(engine.to_numpy(xo), vx)
    xo, _ = engine.scalar_mult_base(cv, ds, flags=OUT_AFFINE, x_only=True)
    assert np.array_equal(engine.to_numpy(xo), bx)
    wx, wy, inf = openssl.double_scalar_mult(cv, u1, k, bx, by, threads=THREADS)
    assert not inf.any()
    rx, ry, fin = engine.double_scalar_mult(cv, du1, dk, dbx, dby)
    assert bool(fin.all())
    assert np.array_equal(engine.to_numpy(rx), wx) and np.array_equal(engine.to_numpy(ry), wy)
