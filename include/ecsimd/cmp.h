// ecsimd/cmp.h -- lane-wise unsigned comparison of two batches (names of the reference's cmp.h:11-51: cmp_lt, cmp_gt,
// cmp_lte, cmp_gte and the four operators).  One device kernel, a < b; the other three are derived from it.
#ifndef ECSIMD_CMP_H
#define ECSIMD_CMP_H
#include <ecsimd/bignum.h>

namespace ecsimd {
template <class BN> hip::mask cmp_lt(wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  hip::mask below(a.size());
  hip::check(ecsimd_hip_cmp_lt(hip::context(), a.data(), b.data(), below.data(), a.size()), "ecsimd_hip_cmp_lt");
  return below;
}
#define ECSIMD_DERIVED_CMP(NAME, OP, EXPR) \
  template <class BN> hip::mask NAME(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return EXPR; } \
  template <class BN> hip::mask operator OP(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return NAME(a, b); }
ECSIMD_DERIVED_CMP(cmp_gt, >, cmp_lt(b, a))
ECSIMD_DERIVED_CMP(cmp_gte, >=, !cmp_lt(a, b))
ECSIMD_DERIVED_CMP(cmp_lte, <=, !cmp_lt(b, a))
#undef ECSIMD_DERIVED_CMP
template <class BN> hip::mask operator<(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return cmp_lt(a, b); }
}  // namespace ecsimd
#endif
