// ecsimd/curve_point.h -- wide_curve_point<Curve>: batch of affine classical points (reference curve_point.h:13-43).
#ifndef ECSIMD_CURVE_POINT_H
#define ECSIMD_CURVE_POINT_H
#include <ecsimd/curve.h>
#include <optional>

namespace ecsimd {
template <class Curve>
struct wide_curve_point {
  using curve_type = Curve;
  using bignum_type = typename Curve::bn_type;
  using WBN = wide_bignum<bignum_type>;
  wide_curve_point() = default;
  wide_curve_point(WBN const& x, WBN const& y) : x_(x), y_(y) {}
  static std::optional<wide_curve_point> from_x(WBN const& x);      // curve_point_ops.h
  WBN const& x() const { return x_; }
  WBN const& y() const { return y_; }
  WBN& x() { return x_; }
  WBN& y() { return y_; }
  size_t size() const { return x_.size(); }
  hip::mask operator==(wide_curve_point const& o) const { return (x() == o.x()) && (y() == o.y()); }
 private:
  WBN x_, y_;
};
}  // namespace ecsimd
#endif
