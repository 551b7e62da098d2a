// ecsimd/sub.h -- sub, sub_if_above (reference sub.h:12-75).
#ifndef ECSIMD_SUB_H
#define ECSIMD_SUB_H
#include <ecsimd/bignum.h>
#include <tuple>

namespace ecsimd {
template <class BN> auto sub(wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  static_assert(BN::nlimbs == 2 || BN::nlimbs == 4, "sub: 128- or 256-bit operands");
  auto r = wide_bignum<BN>::uninitialized(a.size());
  hip::mask borrow(a.size());   // zero-extended 128-bit operands borrow through the upper limbs: same flag
  hip::check(ecsimd_hip_sub(hip::context(), a.data(), b.data(), r.data(), borrow.data(), a.size()), "ecsimd_hip_sub");
  return std::make_tuple(r, borrow);
}
template <class BN> auto sub_no_carry(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return std::get<0>(sub(a, b)); }

// a >= p ? a - p : a    (sub.h:46-69; the optional extra masks of the reference are only used
// internally by mod_add / mod_shift_left_one, which are single kernels here)
template <class BN> wide_bignum<BN> sub_if_above(wide_bignum<BN> const& a, wide_bignum<BN> const& p) {
  auto r = wide_bignum<BN>::uninitialized(a.size());
  hip::check(ecsimd_hip_sub_if_above(hip::context(), a.data(), p.data(), r.data(), a.size()), "ecsimd_hip_sub_if_above");
  return r;
}
}  // namespace ecsimd
#endif
