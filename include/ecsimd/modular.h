// ecsimd/modular.h -- mod_add, mod_sub, mod_shift_left_one (reference modular.h:10-41).
// The reference passes the modulus as a wide; here the prime type P selects the kernel.
#ifndef ECSIMD_MODULAR_H
#define ECSIMD_MODULAR_H
#include <ecsimd/curve.h>

namespace ecsimd {
template <class P, class BN> wide_bignum<BN> mod_add(wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  auto r = wide_bignum<BN>::uninitialized(a.size());
  hip::check(ecsimd_hip_mod_add(hip::context(), hip_field_id<P>(), a.data(), b.data(), r.data(), a.size()), "ecsimd_hip_mod_add"); return r;
}
template <class P, class BN> wide_bignum<BN> mod_sub(wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  auto r = wide_bignum<BN>::uninitialized(a.size());
  hip::check(ecsimd_hip_mod_sub(hip::context(), hip_field_id<P>(), a.data(), b.data(), r.data(), a.size()), "ecsimd_hip_mod_sub"); return r;
}
template <class P, class BN> wide_bignum<BN> mod_shift_left(wide_bignum<BN> const& a, int count) {
  auto r = wide_bignum<BN>::uninitialized(a.size());
  hip::check(ecsimd_hip_mod_shift_left(hip::context(), hip_field_id<P>(), a.data(), count, r.data(), a.size()), "ecsimd_hip_mod_shift_left"); return r;
}
template <class P, class BN> wide_bignum<BN> mod_shift_left_one(wide_bignum<BN> const& a) { return mod_shift_left<P>(a, 1); }
// reference-shaped overloads: the modulus is passed as a wide and must be one of the two primes
template <class BN> wide_bignum<BN> mod_add(wide_bignum<BN> const& a, wide_bignum<BN> const& b, wide_bignum<BN> const& p);
template <class BN> wide_bignum<BN> mod_sub(wide_bignum<BN> const& a, wide_bignum<BN> const& b, wide_bignum<BN> const& p);
template <class BN> wide_bignum<BN> mod_shift_left_one(wide_bignum<BN> const& a, wide_bignum<BN> const& p);
namespace detail {
using p256_tag = p256_prime;
using k256_tag = secp256k1_prime;
inline int curve_of_modulus(bignum_256 const& p) {
  if (p == p256_prime::value) return ECSIMD_HIP_P256;
  if (p == secp256k1_prime::value) return ECSIMD_HIP_SECP256K1;
  throw hip::error("ecsimd: modulus has no HIP kernel (P-256 and secp256k1 only)");
}
}  // namespace detail
template <class BN> wide_bignum<BN> mod_add(wide_bignum<BN> const& a, wide_bignum<BN> const& b, wide_bignum<BN> const& p) {
  return detail::curve_of_modulus(p.get(0)) == ECSIMD_HIP_P256 ? mod_add<detail::p256_tag>(a, b) : mod_add<detail::k256_tag>(a, b);
}
template <class BN> wide_bignum<BN> mod_sub(wide_bignum<BN> const& a, wide_bignum<BN> const& b, wide_bignum<BN> const& p) {
  return detail::curve_of_modulus(p.get(0)) == ECSIMD_HIP_P256 ? mod_sub<detail::p256_tag>(a, b) : mod_sub<detail::k256_tag>(a, b);
}
template <class BN> wide_bignum<BN> mod_shift_left_one(wide_bignum<BN> const& a, wide_bignum<BN> const& p) {
  return detail::curve_of_modulus(p.get(0)) == ECSIMD_HIP_P256 ? mod_shift_left_one<detail::p256_tag>(a) : mod_shift_left_one<detail::k256_tag>(a);
}
}  // namespace ecsimd
#endif
