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
  hip::check(ecsimd_hip_mgry_reduce(hip::context(), hip_curve_id<P>(), a.data(), r.data(), a.size()), "ecsimd_hip_mgry_reduce"); return r;
}
}  // namespace details

template <class P> struct mgry_constants {            // mgry_csts.h:15-24, values from the engine
  static bignum_256 get(int which) { bignum_256 r; hip::check(ecsimd_hip_get_constant(hip_curve_id<P>(), which, r.limbs.data()), "ecsimd_hip_get_constant"); return r; }
  static bignum_256 R_p() { return get(5); }
  static bignum_256 Rsq_p() { return get(6); }
  static bignum_256 Pm1_by_R_p() { return get(7); }
};

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
    hip::check(ecsimd_hip_mgry_from_classical(hip::context(), hip_curve_id<P>(), n.data(), r.data(), n.size()), "ecsimd_hip_mgry_from_classical");
    return wide_mgry_bignum{r};
  }
  WBN to_classical() const {                                                               // mgry.h:52-55
    auto r = WBN::uninitialized(n_.size());
    hip::check(ecsimd_hip_mgry_to_classical(hip::context(), hip_curve_id<P>(), n_.data(), r.data(), n_.size()), "ecsimd_hip_mgry_to_classical");
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
