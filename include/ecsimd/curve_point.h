// ecsimd/curve_point.h -- a batch of affine points in classical coordinates (the wire / test-vector form).
// Keeps the member names of the reference's wide_curve_point (curve_point.h:13-43) -- curve_type, bignum_type,
// x(), y(), operator==, from_x() -- over two device-resident coordinate batches of runtime length, and adds what
// a batch needs beyond a 4-lane register: size(), per-lane decompression and curve membership.
#ifndef ECSIMD_CURVE_POINT_H
#define ECSIMD_CURVE_POINT_H
#include <ecsimd/curve.h>
#include <optional>
#include <utility>

namespace ecsimd {

template <class Curve>
class wide_curve_point {
 public:
  using curve_type  = Curve;
  using bignum_type = typename Curve::bn_type;
  using WBN         = wide_bignum<bignum_type>;

  wide_curve_point() = default;
  wide_curve_point(WBN xs, WBN ys) : coords_{std::move(xs), std::move(ys)} {}

  // ---- decompression (bodies in curve_point_ops.h: they need curve_group)
  // y from x; nullopt unless EVERY lane is on the curve -- the reference's all-or-nothing contract
  static std::optional<wide_curve_point> from_x(WBN const& xs);
  // the same per lane: valid[i] tells whether xs[i] is the abscissa of a curve point (y is meaningless where it is not)
  static wide_curve_point from_x_lanes(WBN const& xs, hip::mask& valid);
  // per lane: y^2 == x^3 + a x + b
  hip::mask on_curve() const;

  WBN&       x()       { return coords_.first; }
  WBN&       y()       { return coords_.second; }
  WBN const& x() const { return coords_.first; }
  WBN const& y() const { return coords_.second; }
  size_t size() const  { return coords_.first.size(); }

  // lane mask: both coordinates equal
  friend hip::mask operator==(wide_curve_point const& l, wide_curve_point const& r) {
    return (l.coords_.first == r.coords_.first) && (l.coords_.second == r.coords_.second);
  }

 private:
  std::pair<WBN, WBN> coords_;     // device-resident batches, shared on copy
};

}  // namespace ecsimd
#endif
