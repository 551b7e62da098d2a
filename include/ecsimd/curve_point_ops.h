// ecsimd/curve_point_ops.h -- out-of-line members of wide_curve_point that need curve_group
// (the reference splits the same way: curve_point_ops.h:12-22).
#ifndef ECSIMD_CURVE_POINT_OPS_H
#define ECSIMD_CURVE_POINT_OPS_H
#include <ecsimd/curve_group.h>

namespace ecsimd {

template <class Curve>
auto wide_curve_point<Curve>::from_x_lanes(WBN const& xs, hip::mask& valid) -> wide_curve_point {
  WBN ys = curve_group<Curve>::compute_y_lanes(xs, valid);         // one kernel: sqrt(x^3 + a x + b) per lane
  return wide_curve_point{xs, std::move(ys)};
}

template <class Curve>
auto wide_curve_point<Curve>::from_x(WBN const& xs) -> std::optional<wide_curve_point> {
  hip::mask valid;
  auto pts = from_x_lanes(xs, valid);
  if (!all(valid)) return std::nullopt;
  return pts;
}

// The root the engine returns for x is one of +-y: a point is on the curve iff its y is that root or its opposite,
// i.e. iff y^2 equals the root's square.  Two Montgomery squarings on the device instead of a second square root.
template <class Curve>
hip::mask wide_curve_point<Curve>::on_curve() const {
  using gfp = GFp<WBN, typename Curve::P>;
  hip::mask has_root;
  const WBN root = curve_group<Curve>::compute_y_lanes(x(), has_root);
  const auto lhs = gfp::from_classical(y()).sqr(), rhs = gfp::from_classical(root).sqr();
  return has_root && (lhs.wbn() == rhs.wbn());
}

}  // namespace ecsimd
#endif
