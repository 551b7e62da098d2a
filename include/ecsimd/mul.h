// ecsimd/mul.h -- mul, square, limb_mul (reference mul.h:150-265).
// On the device these are one Comba pass of v_mad_u64_u32 per element (csrc/field.cuh mul8x8 /
// sqr8); square() returns the exact a*a (the reference's square_u32_zext drops a carry on some
// operands, mul.h:186-190,207 -- see DESIGN.md "Reference defect").
#ifndef ECSIMD_MUL_H
#define ECSIMD_MUL_H
#include <ecsimd/bignum.h>

namespace ecsimd {
namespace detail {
template <class R, class BN> wide_bignum<R> take_low(hip::buffer const& raw8, size_t n) {   // 8-limb device result -> R
  std::vector<uint64_t> h(n * 8); if (n) raw8.download(h.data());
  std::vector<R> o(n);
  for (size_t i = 0; i < n; ++i) for (size_t l = 0; l < R::nlimbs; ++l) o[i].limbs[l] = h[i * 8 + l];
  return wide_bignum<R>(o);
}
}  // namespace detail

// mul.h:63-83 zext_u32x64: the 2N 32-bit digits of every element, each in its own 64-bit limb (the layout the reference's
// AVX2 multiplier works on), and mul.h:85-113 trunc_u64x32, its inverse (each limb's upper half is dropped, as the
// reference's blend does).  Host-side relayouts: the device multiplier needs neither (csrc/field.cuh works on 32-bit words).
template <class BN> auto zext_u32x64(wide_bignum<BN> const& v) {
  static_assert(BN::nlimbs <= 4, "zext_u32x64: at most 256-bit operands");
  using R = bignum<typename BN::limb_type, 2 * BN::nlimbs>;
  auto h = v.host(); std::vector<R> o(h.size());
  for (size_t i = 0; i < h.size(); ++i) for (size_t l = 0; l < BN::nlimbs; ++l) { o[i].limbs[2 * l] = h[i].limbs[l] & 0xffffffffull; o[i].limbs[2 * l + 1] = h[i].limbs[l] >> 32; }
  return wide_bignum<R>(o);
}
template <class BN> auto trunc_u64x32(wide_bignum<BN> const& v) {
  static_assert(BN::nlimbs % 2 == 0, "trunc_u64x32: an even number of limbs");
  using R = bignum<typename BN::limb_type, BN::nlimbs / 2>;
  auto h = v.host(); std::vector<R> o(h.size());
  for (size_t i = 0; i < h.size(); ++i) for (size_t l = 0; l < R::nlimbs; ++l) o[i].limbs[l] = (h[i].limbs[2 * l] & 0xffffffffull) | (h[i].limbs[2 * l + 1] << 32);
  return wide_bignum<R>(o);
}

template <class BN> auto mul(wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  if (a.size() != b.size()) throw hip::error("ecsimd: mul over batches of different length");
  using R = bignum<typename BN::limb_type, 2 * BN::nlimbs>;
  hip::buffer out(a.size() * 8);
  hip::check(ecsimd_hip_mul(hip::context(), a.data(), b.data(), out.data(), a.size()), "ecsimd_hip_mul");
  if constexpr (BN::nlimbs == 4) return wide_bignum<R>::adopt(a.size(), out);
  else return detail::take_low<R, BN>(out, a.size());
}
template <class BN> auto square(wide_bignum<BN> const& a) {
  using R = bignum<typename BN::limb_type, 2 * BN::nlimbs>;
  hip::buffer out(a.size() * 8);
  hip::check(ecsimd_hip_square(hip::context(), a.data(), out.data(), a.size()), "ecsimd_hip_square");
  if constexpr (BN::nlimbs == 4) return wide_bignum<R>::adopt(a.size(), out);
  else return detail::take_low<R, BN>(out, a.size());
}
// bignum x one 32-bit digit per lane (mul.h:252-265; "for testing purposes" in the reference)
template <class BN> auto limb_mul(wide_bignum<BN> const& a, std::vector<uint64_t> const& digit) {
  using R = bignum<typename BN::limb_type, BN::nlimbs + 1>;
  std::vector<BN> d(a.size()); for (size_t i = 0; i < d.size(); ++i) d[i].limbs[0] = digit[i % digit.size()] & 0xffffffffull;
  hip::buffer out(a.size() * 8);
  hip::check(ecsimd_hip_mul(hip::context(), a.data(), wide_bignum<BN>(d).data(), out.data(), a.size()), "ecsimd_hip_mul");
  return detail::take_low<R, BN>(out, a.size());
}
}  // namespace ecsimd
#endif
