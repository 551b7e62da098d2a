// ecsimd/swap.h + ifelse.h counterpart -- lane-masked swap / select (reference swap.h:15-56, ifelse.h:15-49).
#ifndef ECSIMD_SWAP_H
#define ECSIMD_SWAP_H
#include <ecsimd/bignum.h>

namespace ecsimd {
template <class BN> void swap_if(hip::mask const& m, wide_bignum<BN>& a, wide_bignum<BN>& b) {
  if (a.size() != b.size() || m.size() != a.size()) throw hip::error("ecsimd: swap_if over batches of different length");
  a.unshare(); b.unshare();
  hip::check(ecsimd_hip_swap_if(hip::context(), m.data(), a.data(), b.data(), a.size()), "ecsimd_hip_swap_if");
}
// if_else(m, a, b): lanes of a where m, else b (ifelse.h:15-22) -- one select kernel into a fresh batch
template <class BN> wide_bignum<BN> if_else(hip::mask const& m, wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  if (a.size() != b.size() || m.size() != a.size()) throw hip::error("ecsimd: if_else over batches of different length");
  auto r = wide_bignum<BN>::uninitialized(a.size());
  hip::check(ecsimd_hip_if_else(hip::context(), m.data(), a.data(), b.data(), r.data(), a.size()), "ecsimd_hip_if_else");
  return r;
}
}  // namespace ecsimd
#endif
