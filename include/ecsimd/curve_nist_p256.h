// ecsimd/curve_nist_p256.h -- NIST P-256 (SP 800-186), same shape as the reference (curve_nist_p256.h:14-32).
#ifndef ECSIMD_CURVE_NIST_P256_H
#define ECSIMD_CURVE_NIST_P256_H
#include <ecsimd/curve.h>
#include <ecsimd/literals.h>
#include <ecsimd/serialization.h>

namespace ecsimd {
struct curve_nist_p256 {
  using bn_type = bignum_256;
  struct P  { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"ffffffff00000001000000000000000000000000ffffffffffffffffffffffff">()); };
  struct A  { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"ffffffff00000001000000000000000000000000fffffffffffffffffffffffc">()); };
  struct B  { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b">()); };
  struct Gx { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296">()); };
  struct Gy { static constexpr auto value = bn_from_bytes_BE<bn_type>(literals::operator""_hex<"4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5">()); };
};
static_assert(hip_curve_id<curve_nist_p256::P>() == ECSIMD_HIP_P256);
}  // namespace ecsimd
#endif
