// ecsimd/cmp.h -- cmp_lt / gt / lte / gte and the operators (reference cmp.h:11-51).
#ifndef ECSIMD_CMP_H
#define ECSIMD_CMP_H
#include <ecsimd/bignum.h>

namespace ecsimd {
template <class BN> hip::mask cmp_lt(wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  hip::mask m(a.size());
  hip::check(ecsimd_hip_cmp_lt(hip::context(), a.data(), b.data(), m.data(), a.size()), "ecsimd_hip_cmp_lt");
  return m;
}
template <class BN> hip::mask cmp_gt(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return cmp_lt(b, a); }
template <class BN> hip::mask cmp_lte(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return !cmp_gt(a, b); }
template <class BN> hip::mask cmp_gte(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return !cmp_lt(a, b); }
template <class BN> hip::mask operator<(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return cmp_lt(a, b); }
template <class BN> hip::mask operator>(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return cmp_gt(a, b); }
template <class BN> hip::mask operator<=(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return cmp_lte(a, b); }
template <class BN> hip::mask operator>=(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return cmp_gte(a, b); }
}  // namespace ecsimd
#endif
