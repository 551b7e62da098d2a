// ecsimd/mgry.h -- wide_mgry_bignum<WBN, P>: a batch of Montgomery-form residues (reference mgry.h:28-66)
// and details::mgry_reduce<P> (mgry_mul.h:84-121).
#ifndef ECSIMD_MGRY_H
#define ECSIMD_MGRY_H
#include <ecsimd/curve.h>
#include <ecsimd/shift.h>

namespace ecsimd {
namespace details {
// T * 2^-256 mod p for a batch of 512-bit values (mgry_mul.h:84-121)
template <class P> wide_bignum<bignum_256> mgry_reduce(wide_bignum<bignum_512> const& a) {
  auto r = wide_bignum<bignum_256>::uninitialized(a.size());
  hip::check(ecsimd_hip_mgry_reduce(hip::context(), hip_field_id<P>(), a.data(), r.data(), a.size()), "ecsimd_hip_mgry_reduce"); return r;
}
}  // namespace details

template <class P> struct mgry_constants {            // mgry_csts.h:15-24, values from the engine
  static bignum_256 get(int which) { bignum_256 r; hip::check(ecsimd_hip_get_constant(hip_field_id<P>(), which, r.limbs.data()), "ecsimd_hip_get_constant"); return r; }
  static bignum_256 R_p() { return get(5); }
  static bignum_256 Rsq_p() { return get(6); }
  static bignum_256 Pm1_by_R_p() { return get(7); }
};

// mgry.h:18-26 to_mgry<P>(v): v * R mod p at COMPILE TIME (the reference divides with ctbignum; here v mod p bit by bit, then 256 modular
// doublings -- the same residue), for any modulus p >= 2.
namespace details {
constexpr bool bn_geq(bignum_256 const& a, bignum_256 const& b) {
  for (int i = 3; i >= 0; --i) { if (a.limbs[i] != b.limbs[i]) return a.limbs[i] > b.limbs[i]; }
  return true;
}
constexpr bignum_256 bn_sub_wrap(bignum_256 const& a, bignum_256 const& b) {
  bignum_256 r; uint64_t borrow = 0;
  for (int i = 0; i < 4; ++i) { const uint64_t d = a.limbs[i] - b.limbs[i], d2 = d - borrow; borrow = (a.limbs[i] < b.limbs[i]) || (d < borrow); r.limbs[i] = d2; }
  return r;
}
// (2 r + bit) mod p for r < p: 2r + bit < 2p, one subtraction (mod 2^256 when the top bit fell off)
constexpr bignum_256 bn_dbl_mod(bignum_256 const& r, bool bit, bignum_256 const& p) {
  const bool top = r.limbs[3] >> 63;
  bignum_256 d; for (int i = 3; i > 0; --i) d.limbs[i] = (r.limbs[i] << 1) | (r.limbs[i - 1] >> 63);
  d.limbs[0] = (r.limbs[0] << 1) | (bit ? 1u : 0u);
  return (top || bn_geq(d, p)) ? bn_sub_wrap(d, p) : d;
}
}  // namespace details
template <class P> constexpr bignum_256 to_mgry(bignum_256 const& v) {
  constexpr bignum_256 p = P::value;
  bignum_256 r{};
  for (int k = 255; k >= 0; --k) r = details::bn_dbl_mod(r, (v.limbs[k / 64] >> (k % 64)) & 1u, p);      // v mod p
  for (int k = 0; k < 256; ++k) r = details::bn_dbl_mod(r, false, p);                                       // * 2^256
  return r;
}

template <class WBN, class P>
struct wide_mgry_bignum {
  using wide_bignum_type = WBN;
  using bignum_type = typename WBN::value_type;
  using P_type = P;
  using constants_type = mgry_constants<P>;

  wide_mgry_bignum() = default;
  wide_mgry_bignum(WBN const& n) : n_(n) {}
  static wide_mgry_bignum R(size_t lanes = default_lanes) { return wide_mgry_bignum{WBN(lanes, constants_type::R_p())}; }
  static wide_mgry_bignum from_classical(WBN const& n) {                                   // mgry.h:47-50
    auto r = WBN::uninitialized(n.size());
    hip::check(ecsimd_hip_mgry_from_classical(hip::context(), hip_field_id<P>(), n.data(), r.data(), n.size()), "ecsimd_hip_mgry_from_classical");
    return wide_mgry_bignum{r};
  }
  WBN to_classical() const {                                                               // mgry.h:52-55
    auto r = WBN::uninitialized(n_.size());
    hip::check(ecsimd_hip_mgry_to_classical(hip::context(), hip_field_id<P>(), n_.data(), r.data(), n_.size()), "ecsimd_hip_mgry_to_classical");
    return r;
  }
  WBN const& wbn() const { return n_; }
  WBN& wbn() { return n_; }
  size_t size() const { return n_.size(); }
 private:
  WBN n_;
};
}  // namespace ecsimd
#endif
