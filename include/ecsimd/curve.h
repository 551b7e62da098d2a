// ecsimd/curve.h -- curve description (reference curve.h:12-30) and the map from a curve's prime
// to the engine's curve id.  Only the two primes with hand-written kernels are accepted.
#ifndef ECSIMD_CURVE_H
#define ECSIMD_CURVE_H
#include <ecsimd/bignum.h>

namespace ecsimd {
template <class Curve> using curve_bn_t = typename Curve::bn_type;
template <class Curve> using curve_wide_bn_t = wide_bignum<curve_bn_t<Curve>>;

// A 256-bit compile-time constant as a type with a ::value (what the reference's P / A / B / Gx / Gy members are),
// spelled most-significant limb first so that a line reads like the printed hexadecimal number.
template <uint64_t W3, uint64_t W2, uint64_t W1, uint64_t W0>
struct bn256_constant { static constexpr bignum_256 value{{W0, W1, W2, W3}}; };

namespace detail {
using p256_prime = bn256_constant<0xffffffff00000001ull, 0x0000000000000000ull, 0x00000000ffffffffull, 0xffffffffffffffffull>;
using secp256k1_prime = bn256_constant<0xffffffffffffffffull, 0xffffffffffffffffull, 0xffffffffffffffffull, 0xfffffffefffffc2full>;
}
// engine curve id of a prime type P (P::value is the modulus, as in the reference's bignum_cst)
template <class P> constexpr int hip_curve_id() {
  if (P::value == detail::p256_prime::value) return ECSIMD_HIP_P256;
  if (P::value == detail::secp256k1_prime::value) return ECSIMD_HIP_SECP256K1;
  throw "ecsimd: no HIP kernels for this prime (P-256 and secp256k1 only)";
}
}  // namespace ecsimd
#endif
