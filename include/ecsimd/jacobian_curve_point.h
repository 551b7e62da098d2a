// ecsimd/jacobian_curve_point.h -- wide_jacobian_curve_point<Curve>: (X, Y, Z) in Montgomery form,
// x = X/Z^2, y = Y/Z^3 (reference jacobian_curve_point.h:11-68).
#ifndef ECSIMD_JACOBIAN_CURVE_POINT_H
#define ECSIMD_JACOBIAN_CURVE_POINT_H
#include <ecsimd/curve_point.h>
#include <ecsimd/gfp.h>

namespace ecsimd {
template <class Curve>
struct wide_jacobian_curve_point {
  using curve_type = Curve;
  using bignum_type = curve_bn_t<Curve>;
  using WBN = curve_wide_bn_t<Curve>;
  using wide_curve_point_t = wide_curve_point<Curve>;
  using gfp = GFp<WBN, typename Curve::P>;
  static int curve_id() { return hip_curve_id_of<Curve>(); }       // any Curve: the built-in ids, or a run-time registration

  wide_jacobian_curve_point() = default;
  static wide_jacobian_curve_point from_affine(wide_curve_point_t const& pt) {            // :25-31, Z := R mod p
    wide_jacobian_curve_point r; const size_t n = pt.size();
    auto X = WBN::uninitialized(n), Y = WBN::uninitialized(n), Z = WBN::uninitialized(n);
    hip::check(ecsimd_hip_from_affine(hip::context(), curve_id(), pt.x().data(), pt.y().data(), X.data(), Y.data(), Z.data(), n), "ecsimd_hip_from_affine");
    r.x_ = gfp{typename gfp::WMBN{X}}; r.y_ = gfp{typename gfp::WMBN{Y}}; r.z_ = gfp{typename gfp::WMBN{Z}};
    return r;
  }
  wide_curve_point_t to_affine() const {                                                    // :33-42, one inversion per lane
    const size_t n = size(); auto x = WBN::uninitialized(n), y = WBN::uninitialized(n);
    hip::check(ecsimd_hip_to_affine(hip::context(), curve_id(), x_.wbn().data(), y_.wbn().data(), z_.wbn().data(), x.data(), y.data(), n), "ecsimd_hip_to_affine");
    return {x, y};
  }
  hip::mask operator==(wide_jacobian_curve_point const& o) const { return (x().wbn() == o.x().wbn()) && (y().wbn() == o.y().wbn()) && (z().wbn() == o.z().wbn()); }
  wide_jacobian_curve_point opposite() const { wide_jacobian_curve_point r; r.x_ = x_; r.y_ = y_.opposite(); r.z_ = z_; return r; }   // :48-54
  auto& x() { return x_; }
  auto& y() { return y_; }
  auto& z() { return z_; }
  auto const& x() const { return x_; }
  auto const& y() const { return y_; }
  auto const& z() const { return z_; }
  size_t size() const { return x_.size(); }
  void unshare() { x_.wbn().unshare(); y_.wbn().unshare(); z_.wbn().unshare(); }
 private:
  gfp x_, y_, z_;
};
template <class Curve> void swap_if(hip::mask const& m, wide_jacobian_curve_point<Curve>& A, wide_jacobian_curve_point<Curve>& B) {   // swap.h:36-45
  swap_if(m, A.x(), B.x()); swap_if(m, A.y(), B.y()); swap_if(m, A.z(), B.z());
}
template <class Curve> void swap_if_same_z(hip::mask const& m, wide_jacobian_curve_point<Curve>& A, wide_jacobian_curve_point<Curve>& B) {   // swap.h:47-56
  swap_if(m, A.x(), B.x()); swap_if(m, A.y(), B.y());
}
template <class Curve> wide_jacobian_curve_point<Curve> if_else(hip::mask const& m, wide_jacobian_curve_point<Curve> const& a, wide_jacobian_curve_point<Curve> const& b) {
  wide_jacobian_curve_point<Curve> r; r.x() = if_else(m, a.x(), b.x()); r.y() = if_else(m, a.y(), b.y()); r.z() = if_else(m, a.z(), b.z()); return r;
}
}  // namespace ecsimd
#endif
