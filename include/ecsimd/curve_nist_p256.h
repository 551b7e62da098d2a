// ecsimd/curve_nist_p256.h -- NIST P-256 / secp256r1 (FIPS 186-5, SP 800-186 3.2.1.3): y^2 = x^3 - 3x + b over GF(p).
// Member names follow the reference's curve struct (curve_nist_p256.h:14-32): bn_type, P, A, B, Gx, Gy, each
// with a ::value.  The numbers are checked at run time against the engine's own table (ecsimd_hip_get_constant)
// by tests/cpp/host_api_tests.cpp (Curves.ConstantsMatchTheEngine); the engine's table is compared with the
// reference's constants in tests/test_gpu_parity.py.
#ifndef ECSIMD_CURVE_NIST_P256_H
#define ECSIMD_CURVE_NIST_P256_H
#include <ecsimd/curve.h>

namespace ecsimd {
struct curve_nist_p256 {
  using bn_type = bignum_256;
  using P  = detail::p256_prime;                                        // 2^256 - 2^224 + 2^192 + 2^96 - 1
  using A  = bn256_constant<0xffffffff00000001ull, 0x0000000000000000ull, 0x00000000ffffffffull, 0xfffffffffffffffcull>;   // p - 3
  using B  = bn256_constant<0x5ac635d8aa3a93e7ull, 0xb3ebbd55769886bcull, 0x651d06b0cc53b0f6ull, 0x3bce3c3e27d2604bull>;
  using Gx = bn256_constant<0x6b17d1f2e12c4247ull, 0xf8bce6e563a440f2ull, 0x77037d812deb33a0ull, 0xf4a13945d898c296ull>;
  using Gy = bn256_constant<0x4fe342e2fe1a7f9bull, 0x8ee7eb4a7c0f9e16ull, 0x2bce33576b315eceull, 0xcbb6406837bf51f5ull>;
};
static_assert(hip_curve_id<curve_nist_p256::P>() == ECSIMD_HIP_P256);
}  // namespace ecsimd
#endif
