// ecsimd/curve.h -- curve description (reference curve.h:12-30), the map from ANY curve type to the engine's curve id (the two curves with
// hand-written kernels keep their ids; every other Curve is registered with the engine on first use, as the reference's curve_group<Curve>
// instantiates for any Curve) and from ANY odd modulus type to a field id (the field layer is as generic in P as the reference's).
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
// engine curve id of one of the two built-in primes (the table-driven algorithms exist for those curves only)
template <class P> constexpr int hip_curve_id() {
  if (P::value == detail::p256_prime::value) return ECSIMD_HIP_P256;
  if (P::value == detail::secp256k1_prime::value) return ECSIMD_HIP_SECP256K1;
  throw "ecsimd: this entry point exists for P-256 and secp256k1 only";
}
// Engine curve id of ANY curve type with bn_type, P, A, B, Gx, Gy (the reference's concept, curve.h:12-15; p = 3 mod 4 as its GFp needs, gfp.h:84):
// curve_nist_p256 / curve_secp256k1 get their special-form kernels (ids 0 / 1), every other Curve is registered on first use
// (ecsimd_hip_register_curve: host arithmetic only, once per type) and runs on the generic kernels -- points, co-Z formulas, the ladder.  A Curve that
// also names its group order (`using N = ...`, which the reference's concept does not have) gets the generator's comb, double_scalar_mult and ECDSA on top
// of that ladder (another registration, another id: an id's behaviour never changes under its holder).
template <class Curve> inline int hip_curve_id_of() {
  static const int id = [] {
    int cid = -1;
    const uint64_t* order = nullptr;
    if constexpr (requires { Curve::N::value; }) order = Curve::N::value.limbs.data();
    hip::check(ecsimd_hip_register_curve(Curve::P::value.limbs.data(), Curve::A::value.limbs.data(), Curve::B::value.limbs.data(), Curve::Gx::value.limbs.data(),
                                         Curve::Gy::value.limbs.data(), order, 0, &cid), "ecsimd_hip_register_curve");
    return cid;
  }();
  return id;
}
// Field id of a modulus type P for the element-wise field layer (mod_add ... mgry_pow, GFp<WBN, P>): any odd 256-bit P::value, as in the
// reference (mgry_mul.h:84-121 details::mgry_reduce<P>, mgry_csts.h:15-35, gfp.h:17-115).  The two curve primes map to their special-form
// kernels; every other modulus is registered with the engine on first use (ecsimd_hip_register_modulus: host arithmetic only, once per type).
// Prime = the caller's word that P is prime: GFp::inverse then shares one division-step inversion among the lanes instead of raising to P - 2.
template <class P, bool Prime = false> inline int hip_field_id() {
  if constexpr (P::value == detail::p256_prime::value) return ECSIMD_HIP_P256;
  else if constexpr (P::value == detail::secp256k1_prime::value) return ECSIMD_HIP_SECP256K1;
  else {
    static const int id = [] {
      int fid = -1;
      hip::check(ecsimd_hip_register_modulus(P::value.limbs.data(), Prime ? ECSIMD_HIP_MODULUS_PRIME : 0, &fid), "ecsimd_hip_register_modulus");
      return fid;
    }();
    return id;
  }
}
// The group orders (SP 800-186 3.2.1.3, SEC 2 v2 2.4.1) as modulus types: GFp<WBN, p256_order> is arithmetic modulo n (built-in field ids).
using p256_order = bn256_constant<0xffffffff00000000ull, 0xffffffffffffffffull, 0xbce6faada7179e84ull, 0xf3b9cac2fc632551ull>;
using secp256k1_order = bn256_constant<0xffffffffffffffffull, 0xfffffffffffffffeull, 0xbaaedce6af48a03bull, 0xbfd25e8cd0364141ull>;
}  // namespace ecsimd
#endif
