"""CPU suite: algebraic laws of the oracle under random inputs (hypothesis).  These are the
size-independent properties the GPU tests lean on at full batch sizes; checking them on the checker
itself keeps a wrong oracle from vouching for a wrong kernel."""
import numpy as np
from hypothesis import given, settings, strategies as st

from helpers import CURVE_PARAMS, P256, SECP256K1, R, ints_to_arr, arr_to_ints, to_int, ec_add, ec_mul, jacobian_mgry_to_affine_int

u256 = st.integers(min_value=0, max_value=2**256 - 1)
curves = st.sampled_from([P256, SECP256K1])


@settings(max_examples=60, deadline=None)
@given(curves, u256, u256, u256)
def test_field_laws(oracle, cv, a, b, c):
    p = CURVE_PARAMS[cv]["p"]; a %= p; b %= p; c %= p
    A, B, Cc = ints_to_arr([a]), ints_to_arr([b]), ints_to_arr([c])
    val = lambda arr: to_int(arr[0])
    assert val(oracle.mod_add(cv, A, B)) == (a + b) % p and val(oracle.mod_sub(cv, A, B)) == (a - b) % p
    ab = oracle.mgry_mul(cv, A, B)
    assert val(ab) == a * b * pow(R, -1, p) % p
    # distributivity in Montgomery form: (a + b) (x) c == a (x) c + b (x) c
    lhs = oracle.mgry_mul(cv, oracle.mod_add(cv, A, B), Cc)
    rhs = oracle.mod_add(cv, oracle.mgry_mul(cv, A, Cc), oracle.mgry_mul(cv, B, Cc))
    assert np.array_equal(lhs, rhs)
    assert np.array_equal(oracle.mgry_sqr(cv, A), oracle.mgry_mul(cv, A, A))                  # exact-mode square
    assert np.array_equal(oracle.mgry_to_classical(cv, oracle.mgry_from_classical(cv, A)), A)
    if a:
        inv = oracle.gfp_inverse(cv, A)
        assert np.array_equal(oracle.mgry_mul(cv, inv, A), oracle.constants(cv)["r_p"][None])   # a^-1 (x) a == mgry(1)
    assert val(oracle.mod_add(cv, A, oracle.gfp_opposite(cv, A))) == 0


@settings(max_examples=12, deadline=None)
@given(curves, st.integers(min_value=1, max_value=2**255), st.integers(min_value=1, max_value=2**255), st.integers(min_value=2, max_value=2**64))
def test_ladder_is_a_group_homomorphism(oracle, cv, k1, k2, s):
    """(k1 + k2) P == k1 P + k2 P and k1 (k2 P) == (k1 k2) P at the affine level, P = s G."""
    c = CURVE_PARAMS[cv]; n = c["n"]; G = (c["gx"], c["gy"])
    ks = [k1, k2, (k1 + k2) % 2**256, s]
    degenerate = {0, n - 1, (2**256 - n) % n, (2**256 - n - 1) % n}
    if any(k % n in degenerate for k in ks + [k1 * k2 % n]):
        return
    P = ec_mul(cv, s % n, G)
    px, py = ints_to_arr([P[0]] * 3), ints_to_arr([P[1]] * 3)
    J = oracle.scalar_mult(cv, ints_to_arr(ks[:3]), px, py)
    aff = [jacobian_mgry_to_affine_int(cv, *(to_int(v[i]) for v in J)) for i in range(3)]
    assert aff[2] == ec_add(cv, aff[0], aff[1])
    assert aff[0] == ec_mul(cv, k1 % n, P)
    Q = aff[1]                                                                                  # k2 P
    J2 = oracle.scalar_mult(cv, ints_to_arr([k1]), ints_to_arr([Q[0]]), ints_to_arr([Q[1]]))
    assert jacobian_mgry_to_affine_int(cv, *(to_int(v[0]) for v in J2)) == ec_mul(cv, k1 * k2 % n, P)
