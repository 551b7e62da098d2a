// ecsimd/curve_secp256k1.h -- secp256k1 (SEC 2 v2 2.4.1): y^2 = x^3 + 7.  The reference ships no such struct: it
// only uses this prime as a test modulus (tests/mgry.cpp:25-27); BASELINE.json configs[4] swaps
// these parameters into the same 4 x u64 limb path.
#ifndef ECSIMD_CURVE_SECP256K1_H
#define ECSIMD_CURVE_SECP256K1_H
#include <ecsimd/curve.h>

namespace ecsimd {
struct curve_secp256k1 {
  using bn_type = bignum_256;
  using P  = detail::secp256k1_prime;                                   // 2^256 - 2^32 - 977
  using A  = bn256_constant<0, 0, 0, 0>;
  using B  = bn256_constant<0, 0, 0, 7>;
  using Gx = bn256_constant<0x79be667ef9dcbbacull, 0x55a06295ce870b07ull, 0x029bfcdb2dce28d9ull, 0x59f2815b16f81798ull>;
  using Gy = bn256_constant<0x483ada7726a3c465ull, 0x5da4fbfc0e1108a8ull, 0xfd17b448a6855419ull, 0x9c47d08ffb10d4b8ull>;
};
static_assert(hip_curve_id<curve_secp256k1::P>() == ECSIMD_HIP_SECP256K1);
}  // namespace ecsimd
#endif
