// ecsimd/shift.h -- shift_left_one, pad (reference shift.h:13-51).
#ifndef ECSIMD_SHIFT_H
#define ECSIMD_SHIFT_H
#include <ecsimd/add.h>
#include <tuple>

namespace ecsimd {
template <class BN> auto shift_left_one(wide_bignum<BN> const& a) {
  static_assert(BN::nlimbs == 2 || BN::nlimbs == 4, "shift_left_one: 128- or 256-bit operands");
  auto r = wide_bignum<BN>::uninitialized(a.size());
  hip::mask carry(a.size());
  hip::check(ecsimd_hip_shift_left_one(hip::context(), a.data(), r.data(), carry.data(), a.size()), "ecsimd_hip_shift_left_one");
  if constexpr (BN::nlimbs == 2) carry = detail::carry_from_limb(r, 2);
  return std::make_tuple(r, carry);
}
// zero-extend by N limbs (host-side relayout; used for to_classical's 512-bit input)
template <size_t N, class BN> auto pad(wide_bignum<BN> const& v) {
  using R = bignum<typename BN::limb_type, BN::nlimbs + N>;
  auto h = v.host(); std::vector<R> o(h.size());
  for (size_t i = 0; i < h.size(); ++i) for (size_t l = 0; l < BN::nlimbs; ++l) o[i].limbs[l] = h[i].limbs[l];
  return wide_bignum<R>(o);
}
// shift.h:53-79 limb_shift_left<RetLimbs, ShiftBy>: v * 2^(64 ShiftBy) in RetLimbs limbs (what does not fit is dropped);
// shift.h:81-96 limb_shift_right<ShiftBy>: v / 2^(64 ShiftBy) in nlimbs - ShiftBy limbs.  Host-side relayouts like pad().
template <size_t RetLimbs, size_t ShiftBy, class BN> auto limb_shift_left(wide_bignum<BN> const& v) {
  using R = bignum<typename BN::limb_type, RetLimbs>;
  auto h = v.host(); std::vector<R> o(h.size());
  if constexpr (ShiftBy < RetLimbs && ShiftBy < BN::nlimbs)
    for (size_t i = 0; i < h.size(); ++i) for (size_t l = 0; l < BN::nlimbs && l + ShiftBy < RetLimbs; ++l) o[i].limbs[l + ShiftBy] = h[i].limbs[l];
  return wide_bignum<R>(o);
}
template <size_t ShiftBy, class BN> auto limb_shift_right(wide_bignum<BN> const& v) {
  static_assert(ShiftBy < BN::nlimbs, "limb_shift_right: ShiftBy < nlimbs");
  using R = bignum<typename BN::limb_type, BN::nlimbs - ShiftBy>;
  auto h = v.host(); std::vector<R> o(h.size());
  for (size_t i = 0; i < h.size(); ++i) for (size_t l = 0; l < R::nlimbs; ++l) o[i].limbs[l] = h[i].limbs[l + ShiftBy];
  return wide_bignum<R>(o);
}
}  // namespace ecsimd
#endif
