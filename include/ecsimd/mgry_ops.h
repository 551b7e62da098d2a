// ecsimd/mgry_ops.h -- mgry_add/sub/shift_left/mul/sqr/pow and operators (reference mgry_ops.h:10-101).
#ifndef ECSIMD_MGRY_OPS_H
#define ECSIMD_MGRY_OPS_H
#include <ecsimd/mgry.h>
#include <ecsimd/modular.h>

namespace ecsimd {
#define ECSIMD_MGRY_BINOP(NAME, CALL) \
  template <class WBN, class P> wide_mgry_bignum<WBN, P> NAME(wide_mgry_bignum<WBN, P> const& a, wide_mgry_bignum<WBN, P> const& b) { \
    auto r = WBN::uninitialized(a.size()); \
    hip::check(CALL(hip::context(), hip_field_id<P>(), a.wbn().data(), b.wbn().data(), r.data(), a.size()), #CALL); \
    return wide_mgry_bignum<WBN, P>{r}; }
ECSIMD_MGRY_BINOP(mgry_add, ecsimd_hip_mod_add)
ECSIMD_MGRY_BINOP(mgry_sub, ecsimd_hip_mod_sub)
ECSIMD_MGRY_BINOP(mgry_mul, ecsimd_hip_mgry_mul)
#undef ECSIMD_MGRY_BINOP
template <class WBN, class P> wide_mgry_bignum<WBN, P> mgry_sqr(wide_mgry_bignum<WBN, P> const& a) {
  auto r = WBN::uninitialized(a.size());
  hip::check(ecsimd_hip_mgry_sqr(hip::context(), hip_field_id<P>(), a.wbn().data(), r.data(), a.size()), "ecsimd_hip_mgry_sqr");
  return wide_mgry_bignum<WBN, P>{r};
}
template <size_t Count, class WBN, class P> wide_mgry_bignum<WBN, P> mgry_shift_left(wide_mgry_bignum<WBN, P> const& a) {
  static_assert(Count > 0);
  return wide_mgry_bignum<WBN, P>{mod_shift_left<P>(a.wbn(), int(Count))};
}
// a^M for ONE public exponent M (mgry_ops.h:44-86; variable time in M, like the reference)
template <class WBN, class P> wide_mgry_bignum<WBN, P> mgry_pow(wide_mgry_bignum<WBN, P> const& a, typename WBN::value_type const& M) {
  auto r = WBN::uninitialized(a.size());
  hip::check(ecsimd_hip_mgry_pow(hip::context(), hip_field_id<P>(), a.wbn().data(), M.limbs.data(), r.data(), a.size()), "ecsimd_hip_mgry_pow");
  return wide_mgry_bignum<WBN, P>{r};
}
template <class WBN, class P> auto operator+(wide_mgry_bignum<WBN, P> const& a, wide_mgry_bignum<WBN, P> const& b) { return mgry_add(a, b); }
template <class WBN, class P> auto operator-(wide_mgry_bignum<WBN, P> const& a, wide_mgry_bignum<WBN, P> const& b) { return mgry_sub(a, b); }
template <class WBN, class P> auto operator*(wide_mgry_bignum<WBN, P> const& a, wide_mgry_bignum<WBN, P> const& b) { return mgry_mul(a, b); }
}  // namespace ecsimd
#endif
