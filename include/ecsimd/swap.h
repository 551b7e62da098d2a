// ecsimd/swap.h + ifelse.h counterpart -- lane-masked swap / select (reference swap.h:15-56, ifelse.h:15-49).
#ifndef ECSIMD_SWAP_H
#define ECSIMD_SWAP_H
#include <ecsimd/bignum.h>

namespace ecsimd {
template <class BN> void swap_if(hip::mask const& m, wide_bignum<BN>& a, wide_bignum<BN>& b) {
  a.unshare(); b.unshare();
  hip::check(ecsimd_hip_swap_if(hip::context(), m.data(), a.data(), b.data(), a.size()), "ecsimd_hip_swap_if");
}
// if_else(m, a, b): lanes of a where m, else b -- a swap on copies
template <class BN> wide_bignum<BN> if_else(hip::mask const& m, wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  wide_bignum<BN> x = a, y = b;
  swap_if(!m, x, y);
  return x;
}
}  // namespace ecsimd
#endif
