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
}  // namespace ecsimd
#endif
