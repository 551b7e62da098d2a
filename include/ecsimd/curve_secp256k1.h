// ecsimd/curve_secp256k1.h -- secp256k1 (SEC 2 v2 2.4.1).  The reference ships no such struct: it
// only uses this prime as a test modulus (tests/mgry.cpp:25-27); BASELINE.json configs[4] swaps
// these parameters into the same 4 x u64 limb path.
#ifndef ECSIMD_CURVE_SECP256K1_H
#define ECSIMD_CURVE_SECP256K1_H
#include <ecsimd/curve.h>
#include <ecsimd/literals.h>
#include <ecsimd/serialization.h>

namespace ecsimd {
struct curve_secp256k1 {
  using bn_type = bignum_256;
  struct P  { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f">()); };
  struct A  { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"0000000000000000000000000000000000000000000000000000000000000000">()); };
  struct B  { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"0000000000000000000000000000000000000000000000000000000000000007">()); };
  struct Gx { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798">()); };
  struct Gy { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8">()); };
};
static_assert(hip_curve_id<curve_secp256k1::P>() == ECSIMD_HIP_SECP256K1);
}  // namespace ecsimd
#endif
