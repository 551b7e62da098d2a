// ecsimd/curve_point_ops.h -- out-of-line members of wide_curve_point that need curve_group
// (the reference splits the same way: curve_point_ops.h:12-22).
#ifndef ECSIMD_CURVE_POINT_OPS_H
#define ECSIMD_CURVE_POINT_OPS_H
#include <ecsimd/curve_group.h>

namespace ecsimd {

template <class Curve>
auto wide_curve_point<Curve>::from_x(WBN const& xs) -> std::optional<wide_curve_point> {
  hip::mask on_curve;
  WBN ys = curve_group<Curve>::compute_y_lanes(xs, on_curve);      // one kernel: sqrt(x^3 + a x + b) per lane
  if (!all(on_curve)) return std::nullopt;                         // the reference's all-lanes-or-nothing contract
  return wide_curve_point{xs, std::move(ys)};
}

}  // namespace ecsimd
#endif
