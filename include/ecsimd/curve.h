// ecsimd/curve.h -- curve description (reference curve.h:12-30) and the map from a curve's prime
// to the engine's curve id.  Only the two primes with hand-written kernels are accepted.
#ifndef ECSIMD_CURVE_H
#define ECSIMD_CURVE_H
#include <ecsimd/bignum.h>

namespace ecsimd {
template <class Curve> using curve_bn_t = typename Curve::bn_type;
template <class Curve> using curve_wide_bn_t = wide_bignum<curve_bn_t<Curve>>;

namespace detail {
constexpr bignum_256 P256_PRIME{{0xffffffffffffffffull, 0x00000000ffffffffull, 0x0000000000000000ull, 0xffffffff00000001ull}};
constexpr bignum_256 SECP256K1_PRIME{{0xfffffffefffffc2full, 0xffffffffffffffffull, 0xffffffffffffffffull, 0xffffffffffffffffull}};
}
// engine curve id of a prime type P (P::value is the modulus, as in the reference's bignum_cst)
template <class P> constexpr int hip_curve_id() {
  if (P::value == detail::P256_PRIME) return ECSIMD_HIP_P256;
  if (P::value == detail::SECP256K1_PRIME) return ECSIMD_HIP_SECP256K1;
  throw "ecsimd: no HIP kernels for this prime (P-256 and secp256k1 only)";
}
}  // namespace ecsimd
#endif
