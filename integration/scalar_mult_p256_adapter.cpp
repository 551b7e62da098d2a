// scalar_mult_p256_adapter.cpp -- the reference-side binding of INTEGRATION.md section 2, as a real translation unit.
//
// Replaces the reference's lib/scalar_mult_p256.cpp (its one exported function, lib/scalar_mult_p256.cpp:10-12) for a
// maintainer who keeps EVE types in the application: compiled WITH THE REFERENCE'S OWN HEADERS (g++ -std=c++20 -mavx2,
// -I <reference>/include -I <reference>/third-party) and this repo's C ABI (-I <this repo>/include, -lecsimd_hip).
// It converts the reference's AoSoA-4 register layout (u64[limb*4 + lane], eve/arch/cpu/as_register.hpp:55-60) to the
// ABI's AoS layout (u64[elem*4 + limb]) and calls ecsimd_hip_scalar_mult_p256.  tests/test_integration_adapter.py
// compiles and links it where the reference's sources are present (never on the GPU box, never in the product).
#include <ecsimd/curve_group.h>        // the reference's own headers
#include <ecsimd/curve_nist_p256.h>
#include <ecsimd_hip.h>                // this repo's C ABI

#include <cstdlib>

using namespace ecsimd;
using Curve = curve_nist_p256;
using WBN   = curve_wide_bn_t<Curve>;
using WJCP  = wide_jacobian_curve_point<Curve>;

namespace {
ecsimd_hip_ctx* ctx() {
  static ecsimd_hip_ctx* c = [] { ecsimd_hip_ctx* p = nullptr; if (ecsimd_hip_init(0, &p)) std::abort(); return p; }();
  return c;
}
struct dev {
  uint64_t* p = nullptr;
  explicit dev(size_t n) { if (ecsimd_hip_malloc(ctx(), (void**)&p, n * 32)) std::abort(); }
  ~dev() { ecsimd_hip_free(ctx(), p); }
};
void to_aos(uint64_t* dst, WBN const& w) {
  for (int lane = 0; lane < 4; ++lane) { auto c = w.get(lane).cbn(); for (int l = 0; l < 4; ++l) dst[4 * lane + l] = c[l]; }
}
WBN from_aos(const uint64_t* src) {
  return WBN{[&](auto lane, auto) { typename WBN::value_type::cbn_type c; for (int l = 0; l < 4; ++l) c[l] = src[4 * lane + l]; return WBN::value_type::from(c); }};
}
}  // namespace

WJCP scalar_mult_p256(WBN const& x, WJCP const& P) {          // P.z must be mgry(1), as before
  uint64_t h[3][16];
  dev k(4), px(4), py(4), ox(4), oy(4), oz(4);
  to_aos(h[0], x); to_aos(h[1], P.x().wbn()); to_aos(h[2], P.y().wbn());
  ecsimd_hip_memcpy_h2d(ctx(), k.p, h[0], 128); ecsimd_hip_memcpy_h2d(ctx(), px.p, h[1], 128); ecsimd_hip_memcpy_h2d(ctx(), py.p, h[2], 128);
  if (ecsimd_hip_scalar_mult_p256(ctx(), k.p, px.p, py.p, ox.p, oy.p, oz.p, 4)) std::abort();
  ecsimd_hip_memcpy_d2h(ctx(), h[0], ox.p, 128); ecsimd_hip_memcpy_d2h(ctx(), h[1], oy.p, 128); ecsimd_hip_memcpy_d2h(ctx(), h[2], oz.p, 128);
  WJCP r;
  r.x() = WJCP::gfp{from_aos(h[0])}; r.y() = WJCP::gfp{from_aos(h[1])}; r.z() = WJCP::gfp{from_aos(h[2])};
  return r;
}
