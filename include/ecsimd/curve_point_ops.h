// ecsimd/curve_point_ops.h -- wide_curve_point::from_x (reference curve_point_ops.h:12-22).
#ifndef ECSIMD_CURVE_POINT_OPS_H
#define ECSIMD_CURVE_POINT_OPS_H
#include <ecsimd/curve_group.h>

namespace ecsimd {
template <class Curve>
std::optional<wide_curve_point<Curve>> wide_curve_point<Curve>::from_x(typename wide_curve_point<Curve>::WBN const& x) {
  const auto y = curve_group<Curve>::compute_y(x);
  if (!y) return {};
  return {wide_curve_point{x, *y}};
}
}  // namespace ecsimd
#endif
