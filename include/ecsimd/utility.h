// ecsimd/utility.h -- wide_mask_bit (reference utility.h:45-51): bit B of limb L of every lane -> lane mask.
#ifndef ECSIMD_UTILITY_H
#define ECSIMD_UTILITY_H
#include <ecsimd/bignum.h>

namespace ecsimd {
// The reference takes one limb vector (eve::get<L>(x)) and a bit index; here the limb index is explicit.
template <class BN> hip::mask wide_mask_bit(wide_bignum<BN> const& x, size_t limb, size_t bit) {
  hip::mask m(x.size());
  hip::check(ecsimd_hip_mask_bit(hip::context(), x.data(), int(limb * 64 + bit), m.data(), x.size()), "ecsimd_hip_mask_bit");
  return m;
}
// utility.h:36-43 wide_uasr: arithmetic shift right of one limb of every lane (the reference applies it to an
// eve::wide of limbs, eve::get<L>(x); here the limb index is explicit and the result is the host vector of limbs).
// Only wide_mask_bit uses it in the reference; that one is a device kernel here, so this is a host helper.
template <class BN> std::vector<uint64_t> wide_uasr(wide_bignum<BN> const& x, size_t limb, unsigned shift) {
  auto h = x.host(); std::vector<uint64_t> r(h.size());
  for (size_t i = 0; i < h.size(); ++i) {
    const uint64_t v = h[i].limbs[limb];
    r[i] = shift >= 64 ? (v >> 63 ? ~0ull : 0ull) : (v >> shift) | ((v >> 63) && shift ? ~0ull << (64 - shift) : 0ull);
  }
  return r;
}
}  // namespace ecsimd
#endif
