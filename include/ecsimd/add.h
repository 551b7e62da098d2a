// ecsimd/add.h -- add (reference add.h:11-41).  Returns {sum mod 2^bits, carry mask}.
#ifndef ECSIMD_ADD_H
#define ECSIMD_ADD_H
#include <ecsimd/bignum.h>
#include <tuple>

namespace ecsimd {
namespace detail {
// 128-bit values travel zero-extended in 256-bit device elements: the 128-bit carry is limb 2.
template <class WBN> hip::mask carry_from_limb(WBN const& r256like, size_t limb) {
  std::vector<uint64_t> raw(r256like.size() * WBN::dev_limbs);
  if (r256like.size()) hip::check(ecsimd_hip_memcpy_d2h(hip::context(), raw.data(), r256like.data(), raw.size() * 8), "d2h");
  std::vector<uint8_t> h(r256like.size());
  for (size_t i = 0; i < h.size(); ++i) h[i] = raw[i * WBN::dev_limbs + limb] & 1;
  hip::mask m(h.size());
  if (!h.empty()) hip::check(ecsimd_hip_memcpy_h2d(hip::context(), m.data(), h.data(), h.size()), "h2d");
  return m;
}
}  // namespace detail

template <class BN> auto add(wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  static_assert(BN::nlimbs == 2 || BN::nlimbs == 4, "add: 128- or 256-bit operands");
  auto r = wide_bignum<BN>::uninitialized(a.size());
  hip::mask carry(a.size());
  hip::check(ecsimd_hip_add(hip::context(), a.data(), b.data(), r.data(), carry.data(), a.size()), "ecsimd_hip_add");
  if constexpr (BN::nlimbs == 2) carry = detail::carry_from_limb(r, 2);
  return std::make_tuple(r, carry);
}
template <class BN> auto add_no_carry(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return std::get<0>(add(a, b)); }
}  // namespace ecsimd
#endif
