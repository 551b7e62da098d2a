// ecsimd/bignum.h -- bignum<Limb, N> and its batched ("wide") form.
// Mirrors the reference's include/ecsimd/bignum.h: bignum (38-95), bignum_128/256/512 (97-99),
// wide_bignum (101-102), cmp_res_t (136-137), bn_limb_t / bn_nlimbs (129-133).
//
// Difference by design: the reference's wide_bignum is eve::wide<BN, fixed<4>> (4 lanes in ymm
// registers).  Here a wide_bignum is a DEVICE-RESIDENT batch of runtime length n; n = 4 (the
// default of every constructor the reference's tests use) reproduces one eve::wide.
#ifndef ECSIMD_BIGNUM_H
#define ECSIMD_BIGNUM_H

#include <ecsimd/hip_runtime.h>

#include <array>
#include <cstddef>
#include <cstdint>
#include <type_traits>
#include <vector>

namespace ecsimd {

constexpr size_t default_lanes = 4;   // the reference's eve::fixed<4>

template <class LimbType, size_t NLimbs>
struct bignum {
  static_assert(std::is_same_v<LimbType, uint64_t>, "the HIP engine works on 64-bit limbs");
  using limb_type = LimbType;
  static constexpr size_t nlimbs = NLimbs;
  std::array<LimbType, NLimbs> limbs{};   // little-endian limb order, like the reference's tuple

  // bignum.h:61-67 cbn() / from(): the limbs as the plain array ctbignum's big_int is (the reference bit_casts between the two)
  using cbn_type = std::array<LimbType, NLimbs>;
  constexpr cbn_type cbn() const { return limbs; }
  static constexpr bignum from(cbn_type const& v) { bignum r; r.limbs = v; return r; }
  static constexpr bignum from(limb_type v0) { bignum r; r.limbs[0] = v0; return r; }
  constexpr limb_type operator[](size_t i) const { return limbs[i]; }
  constexpr limb_type& operator[](size_t i) { return limbs[i]; }
  friend constexpr bool operator==(bignum const& a, bignum const& b) { return a.limbs == b.limbs; }
  friend constexpr bool operator!=(bignum const& a, bignum const& b) { return !(a == b); }
};
template <size_t I, class L, size_t N> constexpr L get(bignum<L, N> const& b) { return b.limbs[I]; }

using bignum_128 = bignum<uint64_t, 2>;
using bignum_256 = bignum<uint64_t, 4>;
using bignum_512 = bignum<uint64_t, 8>;

template <class T> struct bn_traits;
template <class L, size_t N> struct bn_traits<bignum<L, N>> { using bignum_type = bignum<L, N>; using limb_type = L; static constexpr size_t nlimbs = N; };
template <class T> using bn_limb_t = typename bn_traits<T>::limb_type;
template <class T> inline constexpr size_t bn_nlimbs = bn_traits<T>::nlimbs;
template <class T> using bn_t = typename bn_traits<T>::bignum_type;

template <class Bignum>
class wide_bignum {
 public:
  using value_type = Bignum;
  using limb_type = typename Bignum::limb_type;
  static constexpr size_t nlimbs = Bignum::nlimbs;
  // The device kernels take 256-bit (4-limb) and 512-bit (8-limb) elements; 128-bit values are
  // carried zero-extended in 256-bit elements (the 128-bit type only appears in the reference's
  // unit tests, tests/ops.cpp).
  static constexpr size_t dev_limbs = nlimbs <= 4 ? 4 : 8;

  wide_bignum() = default;
  explicit wide_bignum(Bignum const& v) : wide_bignum(default_lanes, v) {}                 // splat, 4 lanes
  wide_bignum(size_t n, Bignum const& v) : n_(n), buf_(n * dev_limbs) { std::vector<Bignum> h(n, v); upload(h); }
  explicit wide_bignum(std::vector<Bignum> const& lanes) : n_(lanes.size()), buf_(lanes.size() * dev_limbs) { upload(lanes); }
  template <class Gen, class = std::enable_if_t<std::is_invocable_v<Gen, size_t, size_t>>>
  explicit wide_bignum(Gen&& g) : wide_bignum(default_lanes, std::forward<Gen>(g)) {}      // per-lane generator, 4 lanes
  template <class Gen, class = std::enable_if_t<std::is_invocable_v<Gen, size_t, size_t>>>
  wide_bignum(size_t n, Gen&& g) : n_(n), buf_(n * dev_limbs) {
    std::vector<Bignum> h; for (size_t i = 0; i < n; ++i) h.push_back(g(i, n)); upload(h);
  }
  // adopt device memory produced by a kernel
  static wide_bignum adopt(size_t n, hip::buffer b) { wide_bignum w; w.n_ = n; w.buf_ = std::move(b); return w; }
  static wide_bignum uninitialized(size_t n) { return adopt(n, hip::buffer(n * dev_limbs)); }

  size_t size() const { return n_; }
  uint64_t* data() const { return buf_.data(); }
  std::vector<Bignum> host() const {
    std::vector<uint64_t> raw(n_ * dev_limbs); if (n_) buf_.download(raw.data());
    std::vector<Bignum> out(n_);
    for (size_t i = 0; i < n_; ++i) for (size_t l = 0; l < nlimbs; ++l) out[i].limbs[l] = raw[i * dev_limbs + l];
    return out;
  }
  Bignum get(size_t lane) const { return host().at(lane); }
  // in-place kernels (the co-Z "update" parameters) must not write through a shared buffer
  void unshare() { if (buf_.shared()) buf_ = buf_.clone(); }

 private:
  void upload(std::vector<Bignum> const& h) {
    std::vector<uint64_t> raw(n_ * dev_limbs, 0);
    for (size_t i = 0; i < n_; ++i) for (size_t l = 0; l < nlimbs; ++l) raw[i * dev_limbs + l] = h[i].limbs[l];
    if (n_) buf_.upload(raw.data());
  }
  size_t n_ = 0;
  hip::buffer buf_;
};
template <class BN> struct bn_traits<wide_bignum<BN>> : bn_traits<BN> {};

template <class WBN> using cmp_res_t = hip::mask;                                           // bignum.h:136-137
template <class WBN> using wbn_zext_t = wide_bignum<bignum<bn_limb_t<WBN>, bn_nlimbs<WBN> * 2>>;   // bignum.h:166-167

// lane-wise equality of two wides (the reference gets it from eve's product-type ==)
template <class BN> hip::mask operator==(wide_bignum<BN> const& a, wide_bignum<BN> const& b) {
  if (a.size() != b.size()) throw hip::error("ecsimd: comparing batches of different length");
  hip::mask m(a.size());
  hip::check(ecsimd_hip_cmp_eq(hip::context(), a.data(), b.data(), (int)wide_bignum<BN>::nlimbs, m.data(), a.size()), "ecsimd_hip_cmp_eq");
  return m;
}
template <class BN> hip::mask operator!=(wide_bignum<BN> const& a, wide_bignum<BN> const& b) { return !(a == b); }

}  // namespace ecsimd
#endif
