// ecsimd/curve_point.h -- a batch of affine points in classical coordinates.
// API of the reference's wide_curve_point (curve_point.h:13-43): x(), y(), ==, from_x().
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

  // Decompression: y from x, nullopt unless EVERY lane is on the curve (defined in curve_point_ops.h,
  // which needs curve_group; per-lane validity: curve_group<Curve>::compute_y_lanes / sec1_decode).
  static std::optional<wide_curve_point> from_x(WBN const& xs);

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
