// ecsimd/sec1.h -- batch wire formats on the device: big-endian scalars / field elements
// (the batch form of serialization.h:12-48) and SEC 1 v2 2.3.3-2.3.4 points.  Not in the reference
// (SURVEY.md 8(f) rank 3); the host-side single-value functions stay in <ecsimd/serialization.h>.
#ifndef ECSIMD_SEC1_H
#define ECSIMD_SEC1_H
#include <ecsimd/curve_point.h>
#include <vector>

namespace ecsimd {
namespace detail {
struct dev_bytes {
  uint8_t* p = nullptr; size_t n = 0;
  explicit dev_bytes(size_t bytes) : n(bytes) { hip::check(ecsimd_hip_malloc(hip::context(), (void**)&p, bytes), "ecsimd_hip_malloc"); }
  ~dev_bytes() { ecsimd_hip_free(hip::context(), p); }
  dev_bytes(dev_bytes const&) = delete;
};
}  // namespace detail

// n x 32 big-endian bytes -> batch
inline wide_bignum<bignum_256> wide_from_bytes_BE(std::vector<uint8_t> const& bytes) {
  const size_t n = bytes.size() / 32; detail::dev_bytes d(bytes.size());
  hip::check(ecsimd_hip_memcpy_h2d(hip::context(), d.p, bytes.data(), bytes.size()), "h2d");
  auto r = wide_bignum<bignum_256>::uninitialized(n);
  hip::check(ecsimd_hip_from_bytes_be(hip::context(), d.p, r.data(), n), "ecsimd_hip_from_bytes_be");
  hip::sync(); return r;
}
inline std::vector<uint8_t> wide_to_bytes_BE(wide_bignum<bignum_256> const& v) {
  detail::dev_bytes d(v.size() * 32); std::vector<uint8_t> out(v.size() * 32);
  hip::check(ecsimd_hip_to_bytes_be(hip::context(), v.data(), d.p, v.size()), "ecsimd_hip_to_bytes_be");
  hip::check(ecsimd_hip_memcpy_d2h(hip::context(), out.data(), d.p, out.size()), "d2h"); return out;
}
template <class Curve> std::vector<uint8_t> sec1_encode(wide_curve_point<Curve> const& pts, bool compressed) {
  const size_t rec = compressed ? 33 : 65; detail::dev_bytes d(pts.size() * rec); std::vector<uint8_t> out(pts.size() * rec);
  hip::check(ecsimd_hip_sec1_encode(hip::context(), hip_curve_id_of<Curve>(), pts.x().data(), pts.y().data(), d.p, pts.size(), compressed), "ecsimd_hip_sec1_encode");
  hip::check(ecsimd_hip_memcpy_d2h(hip::context(), out.data(), d.p, out.size()), "d2h"); return out;
}
template <class Curve> wide_curve_point<Curve> sec1_decode(std::vector<uint8_t> const& bytes, bool compressed, hip::mask& ok) {
  const size_t rec = compressed ? 33 : 65, n = bytes.size() / rec; detail::dev_bytes d(bytes.size());
  hip::check(ecsimd_hip_memcpy_h2d(hip::context(), d.p, bytes.data(), bytes.size()), "h2d");
  using WBN = typename wide_curve_point<Curve>::WBN;
  auto x = WBN::uninitialized(n), y = WBN::uninitialized(n); ok = hip::mask(n);
  hip::check(ecsimd_hip_sec1_decode(hip::context(), hip_curve_id_of<Curve>(), d.p, x.data(), y.data(), ok.data(), n, compressed), "ecsimd_hip_sec1_decode");
  hip::sync(); return {x, y};
}
}  // namespace ecsimd
#endif
