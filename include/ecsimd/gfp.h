// ecsimd/gfp.h -- GFp<WBN, P>: field element batch in Montgomery form (reference gfp.h:17-115).
#ifndef ECSIMD_GFP_H
#define ECSIMD_GFP_H
#include <ecsimd/mgry_ops.h>
#include <optional>

namespace ecsimd {
template <class WBN_, class P>
struct GFp {
  using P_type = P;
  using WBN = WBN_;
  using BN = typename WBN::value_type;
  using WMBN = wide_mgry_bignum<WBN, P>;

  GFp() = default;
  GFp(WMBN const& n) : n_(n) {}
  static GFp one(size_t lanes = default_lanes) { return GFp{WMBN::R(lanes)}; }
  static GFp from_classical(WBN const& n) { return {WMBN::from_classical(n)}; }
  WBN to_classical() const { return n_.to_classical(); }
  GFp inverse() const {                                                       // gfp.h:42-44: x^(p-2)
    auto r = WBN::uninitialized(n_.size());
    hip::check(ecsimd_hip_gfp_inverse(hip::context(), hip_field_id<P>(), wbn().data(), r.data(), r.size()), "ecsimd_hip_gfp_inverse");
    return GFp{WMBN{r}};
  }
  // gfp.h:46-54: x^((p+1)/4); like the reference, nullopt if ANY lane has no square root.
  // sqrt_lanes() additionally reports validity per lane.
  std::optional<GFp> sqrt() const { hip::mask ok; GFp r = sqrt_lanes(ok); if (!all(ok)) return {}; return {r}; }
  GFp sqrt_lanes(hip::mask& ok) const {
    auto r = WBN::uninitialized(n_.size()); ok = hip::mask(n_.size());
    hip::check(ecsimd_hip_gfp_sqrt(hip::context(), hip_field_id<P>(), wbn().data(), r.data(), ok.data(), r.size()), "ecsimd_hip_gfp_sqrt");
    return GFp{WMBN{r}};
  }
  GFp sqr() const { return {mgry_sqr(n_)}; }
  GFp opposite() const {                                                      // gfp.h:60-64
    auto r = WBN::uninitialized(n_.size());
    hip::check(ecsimd_hip_gfp_opposite(hip::context(), hip_field_id<P>(), wbn().data(), r.data(), r.size()), "ecsimd_hip_gfp_opposite");
    return GFp{WMBN{r}};
  }
  auto const& wbn() const { return n_.wbn(); }
  auto& wbn() { return n_.wbn(); }
  auto const& wmbn() const { return n_; }
  auto& wmbn() { return n_; }
  size_t size() const { return n_.size(); }
 private:
  WMBN n_;
};
template <class WBN, class P> GFp<WBN, P> operator+(GFp<WBN, P> const& a, GFp<WBN, P> const& b) { return {mgry_add(a.wmbn(), b.wmbn())}; }
template <class WBN, class P> GFp<WBN, P> operator-(GFp<WBN, P> const& a, GFp<WBN, P> const& b) { return {mgry_sub(a.wmbn(), b.wmbn())}; }
template <class WBN, class P> GFp<WBN, P> operator*(GFp<WBN, P> const& a, GFp<WBN, P> const& b) { return {mgry_mul(a.wmbn(), b.wmbn())}; }
template <size_t Count, class WBN, class P> GFp<WBN, P> gfp_shift_left(GFp<WBN, P> const& a) { return {mgry_shift_left<Count>(a.wmbn())}; }
template <class WBN, class P> void swap_if(hip::mask const& m, GFp<WBN, P>& a, GFp<WBN, P>& b) { swap_if(m, a.wbn(), b.wbn()); }
template <class WBN, class P> GFp<WBN, P> if_else(hip::mask const& m, GFp<WBN, P> const& a, GFp<WBN, P> const& b) {
  return GFp<WBN, P>{wide_mgry_bignum<WBN, P>{if_else(m, a.wbn(), b.wbn())}};
}
}  // namespace ecsimd
#endif
