// scalar_mult_p256_adapter.cpp -- the reference-side binding of INTEGRATION.md section 2, as a real translation unit.
//
// Replaces the reference's lib/scalar_mult_p256.cpp (its one exported function, lib/scalar_mult_p256.cpp:10-12) for a
// maintainer who keeps EVE types in the application: compiled WITH THE REFERENCE'S OWN HEADERS (g++ -std=c++20 -mavx2,
// -I <reference>/include -I <reference>/third-party) and this repo's C ABI (-I <this repo>/include, -lecsimd_hip).
// It converts the reference's AoSoA-4 register layout (u64[limb*4 + lane], eve/arch/cpu/as_register.hpp:55-60) to the
// ABI's AoS layout (u64[elem*4 + limb]) and calls ecsimd_hip_scalar_mult_p256 -- ONE launch for a whole span of wides
// (the batch form; the four-lane signature of the reference is the span of length one).
// EXECUTED: oracle/adapter_driver.cpp links this file, replays the reference's ScalarMult scenarios (tests/curve_group.cpp:117-173)
// and compares lane-distinct wides limb for limb with curve_group<curve_nist_p256>::scalar_mult computed by the reference in the
// same process; tests/test_integration_adapter.py runs that binary on the GPU.
#include "scalar_mult_p256_adapter.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ecsimd;
using namespace ecsimd_mi355x;

namespace {
[[noreturn]] void die(const char* what, int rc, ecsimd_hip_ctx* c) {
  std::fprintf(stderr, "scalar_mult_p256 adapter: %s failed (%d): %s\n", what, rc, c ? ecsimd_hip_last_error(c) : "");
  std::abort();                                              // the reference's signature has no error channel
}
ecsimd_hip_ctx* ctx() {
  static ecsimd_hip_ctx* c = [] { ecsimd_hip_ctx* p = nullptr; if (int rc = ecsimd_hip_init(0, &p)) die("ecsimd_hip_init", rc, nullptr); return p; }();
  return c;
}
// grow-only device staging: k, px, py in; ox, oy, oz out (n elements of 32 bytes each)
struct staging {
  uint64_t* dev = nullptr; size_t elems = 0;
  uint64_t* get(size_t n) {
    if (n > elems) {
      if (dev) ecsimd_hip_free(ctx(), dev);
      if (int rc = ecsimd_hip_malloc(ctx(), (void**)&dev, 6 * n * 32)) die("ecsimd_hip_malloc", rc, ctx());
      elems = n;
    }
    return dev;
  }
};
// wide w (lanes 0..3) -> elements 4w .. 4w+3 of an AoS array: element e, limb l at dst[4 * e + l]
void to_aos(uint64_t* dst, WBN const& w) {
  for (int lane = 0; lane < 4; ++lane) { auto c = w.get(lane).cbn(); for (int l = 0; l < 4; ++l) dst[4 * lane + l] = c[l]; }
}
WBN from_aos(const uint64_t* src) {
  return WBN{[&](auto lane, auto) { typename WBN::value_type::cbn_type c; for (int l = 0; l < 4; ++l) c[l] = src[4 * lane + l]; return WBN::value_type::from(c); }};
}
}  // namespace

ecsimd_hip_ctx* scalar_mult_p256_context() { return ctx(); }

void scalar_mult_p256(std::span<const WBN> x, std::span<const WJCP> P, std::span<WJCP> out) {
  const size_t wides = x.size(), n = 4 * wides;
  if (P.size() != wides || out.size() != wides) die("span lengths", -1, nullptr);
  if (!n) return;
  static staging st;                                         // one caller thread, like the context (include/ecsimd_hip.h)
  static std::vector<uint64_t> host;
  host.resize(3 * n * 4);
  uint64_t* hk = host.data(); uint64_t* hx = hk + 4 * n; uint64_t* hy = hx + 4 * n;
  for (size_t w = 0; w < wides; ++w) { to_aos(hk + 16 * w, x[w]); to_aos(hx + 16 * w, P[w].x().wbn()); to_aos(hy + 16 * w, P[w].y().wbn()); }
  uint64_t* d = st.get(n);
  uint64_t* dk = d; uint64_t* dx = d + 4 * n; uint64_t* dy = dx + 4 * n; uint64_t* ox = dy + 4 * n; uint64_t* oy = ox + 4 * n; uint64_t* oz = oy + 4 * n;
  if (int rc = ecsimd_hip_memcpy_h2d(ctx(), dk, hk, 3 * n * 32)) die("h2d", rc, ctx());               // k, px, py are contiguous on both sides
  if (int rc = ecsimd_hip_scalar_mult_p256(ctx(), dk, dx, dy, ox, oy, oz, n)) die("ecsimd_hip_scalar_mult_p256", rc, ctx());
  if (int rc = ecsimd_hip_memcpy_d2h(ctx(), hk, ox, 3 * n * 32)) die("d2h", rc, ctx());               // (synchronises the stream)
  for (size_t w = 0; w < wides; ++w) {
    WJCP r;
    r.x() = WJCP::gfp{from_aos(hk + 16 * w)}; r.y() = WJCP::gfp{from_aos(hx + 16 * w)}; r.z() = WJCP::gfp{from_aos(hy + 16 * w)};
    out[w] = r;
  }
}

WJCP scalar_mult_p256(WBN const& x, WJCP const& P) {          // P.z must be mgry(1), as before
  WJCP r;
  scalar_mult_p256(std::span<const WBN>(&x, 1), std::span<const WJCP>(&P, 1), std::span<WJCP>(&r, 1));
  return r;
}
