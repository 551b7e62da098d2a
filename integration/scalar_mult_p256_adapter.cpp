// scalar_mult_p256_adapter.cpp -- the reference-side binding of INTEGRATION.md section 2, as a real translation unit.
//
// Replaces the reference's lib/scalar_mult_p256.cpp (its one exported function, lib/scalar_mult_p256.cpp:10-12) for a
// maintainer who keeps EVE types in the application: compiled WITH THE REFERENCE'S OWN HEADERS (g++ -std=c++20 -mavx2,
// -I <reference>/include -I <reference>/third-party) and this repo's C ABI (-I <this repo>/include, -lecsimd_hip).
// It converts the reference's AoSoA-4 register layout (u64[limb*4 + lane], eve/arch/cpu/as_register.hpp:55-60) to the
// ABI's AoS layout (u64[elem*4 + limb]) and calls ecsimd_hip_scalar_mult_p256 -- ONE launch for a whole span of wides
// (the batch form; the four-lane signature of the reference is the span of length one).  The conversion runs ON THE DEVICE
// (ecsimd_hip_wide4_to_lanes / _lanes_to_wide4: the spans are copied as they are) wherever the compiler laid the reference's
// types out as their definitions say -- checked at first use; the per-lane host loop is the other path.
// EXECUTED: oracle/adapter_driver.cpp links this file, replays the reference's ScalarMult scenarios (tests/curve_group.cpp:117-173)
// and compares lane-distinct wides limb for limb with curve_group<curve_nist_p256>::scalar_mult computed by the reference in the
// same process; tests/test_integration_adapter.py runs that binary on the GPU.
#include "scalar_mult_p256_adapter.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
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
// The reference's types as bytes.  A wide_bignum is eve::wide<bignum, fixed<4>>: limb-major, u64[limb * 4 + lane], 128 bytes (bignum.h:99-100;
// eve/arch/cpu/as_register.hpp:55-60), and a Jacobian point holds three of them.  Where that is what this compiler laid out -- CHECKED here on a test
// pattern and on the member addresses of a real object, once -- the spans go to the device as they are and the 4 x 4 transposition runs there
// (ecsimd_hip_wide4_to_lanes / _lanes_to_wide4); otherwise the per-lane conversion above does it on the host.
struct raw_layout { bool ok = false; size_t rec = 0, off[3] = {0, 0, 0}; };
raw_layout probe_layout() {
  raw_layout L;
  if constexpr (sizeof(WBN) == 128 && std::is_trivially_copyable_v<WBN> && std::is_trivially_copyable_v<WJCP> && sizeof(WJCP) % 8 == 0) {
    const WBN w{[](auto lane, auto) { typename WBN::value_type::cbn_type c; for (int l = 0; l < 4; ++l) c[l] = 0x0101010101010101ull * (uint64_t)(16 * lane + l + 1); return WBN::value_type::from(c); }};
    uint64_t raw[16];
    std::memcpy(raw, &w, sizeof raw);
    for (int l = 0; l < 4; ++l) for (int lane = 0; lane < 4; ++lane) if (raw[4 * l + lane] != 0x0101010101010101ull * (uint64_t)(16 * lane + l + 1)) return L;
    WJCP P;
    const char* base = reinterpret_cast<const char*>(&P);
    const char* m[3] = {reinterpret_cast<const char*>(&P.x().wbn()), reinterpret_cast<const char*>(&P.y().wbn()), reinterpret_cast<const char*>(&P.z().wbn())};
    for (int i = 0; i < 3; ++i) {
      if (m[i] < base || m[i] + 128 > base + sizeof(WJCP) || ((m[i] - base) & 7)) return L;
      L.off[i] = (size_t)(m[i] - base);
    }
    L.rec = sizeof(WJCP);
    L.ok = std::getenv("ECSIMD_ADAPTER_HOST_TRANSPOSE") == nullptr;      // (the A/B switch of oracle/adapter_driver.cpp)
  }
  return L;
}
const raw_layout& layout() { static const raw_layout L = probe_layout(); return L; }

// Two contexts (two HIP streams) and two staging blocks: chunk c runs on side c & 1.  While the ladder of one chunk runs, the host thread copies the next
// chunk in on the other side and launches it, then waits for the previous chunk and copies it out -- the copies of one chunk overlap the ladder of its
// neighbour, the GPU always has a launch queued, and the caller's spans stay ordinary pageable memory.  One chunk = 2^19 lanes (where the ladder's rate
// saturates); a batch of up to one chunk is a single pass.
constexpr size_t CHUNK_WIDES = (size_t)1 << 17;
struct side {
  ecsimd_hip_ctx* c = nullptr; uint8_t* dev = nullptr; size_t bytes = 0;
  uint8_t* get(size_t need) {
    if (need > bytes) {
      if (dev) ecsimd_hip_free(c, dev);
      if (int rc = ecsimd_hip_malloc(c, (void**)&dev, need)) die("ecsimd_hip_malloc", rc, c);
      bytes = need;
    }
    return dev;
  }
};
side& side_of(int i) {
  static side s[2];
  if (!s[i].c) { if (i == 0) s[0].c = ctx(); else if (int rc = ecsimd_hip_init(0, &s[1].c)) die("ecsimd_hip_init", rc, nullptr); }
  return s[i];
}
struct chunk_plan { uint8_t *dxw, *dpw, *dow; uint64_t *dk, *px, *py, *ox, *oy, *oz; size_t pb; };
chunk_plan plan(side& S, size_t wides, size_t rec) {
  const size_t n = 4 * wides, xb = wides * 128, pb = wides * rec;
  uint8_t* d = S.get(xb + 2 * pb + 6 * n * 32);
  chunk_plan P;
  P.dxw = d; P.dpw = d + xb; P.dow = P.dpw + pb; P.pb = pb;
  P.dk = reinterpret_cast<uint64_t*>(P.dow + pb);
  P.px = P.dk + 4 * n; P.py = P.px + 4 * n; P.ox = P.py + 4 * n; P.oy = P.ox + 4 * n; P.oz = P.oy + 4 * n;
  return P;
}
void launch_chunk(side& S, const chunk_plan& B, const WBN* x, const WJCP* P, size_t wides) {
  const raw_layout& L = layout();
  ecsimd_hip_ctx* c = S.c;
  if (int rc = ecsimd_hip_memcpy_h2d(c, B.dxw, x, wides * 128)) die("h2d", rc, c);
  if (int rc = ecsimd_hip_memcpy_h2d(c, B.dpw, P, B.pb)) die("h2d", rc, c);
  if (int rc = ecsimd_hip_wide4_to_lanes(c, B.dxw, 128, 0, B.dk, wides)) die("ecsimd_hip_wide4_to_lanes", rc, c);
  if (int rc = ecsimd_hip_wide4_to_lanes(c, B.dpw, L.rec, L.off[0], B.px, wides)) die("ecsimd_hip_wide4_to_lanes", rc, c);
  if (int rc = ecsimd_hip_wide4_to_lanes(c, B.dpw, L.rec, L.off[1], B.py, wides)) die("ecsimd_hip_wide4_to_lanes", rc, c);
  if (int rc = ecsimd_hip_scalar_mult_p256(c, B.dk, B.px, B.py, B.ox, B.oy, B.oz, 4 * wides)) die("ecsimd_hip_scalar_mult_p256", rc, c);
  if (int rc = ecsimd_hip_lanes_to_wide4(c, B.ox, B.dow, L.rec, L.off[0], wides)) die("ecsimd_hip_lanes_to_wide4", rc, c);
  if (int rc = ecsimd_hip_lanes_to_wide4(c, B.oy, B.dow, L.rec, L.off[1], wides)) die("ecsimd_hip_lanes_to_wide4", rc, c);
  if (int rc = ecsimd_hip_lanes_to_wide4(c, B.oz, B.dow, L.rec, L.off[2], wides)) die("ecsimd_hip_lanes_to_wide4", rc, c);
}
void drain_chunk(side& S, const chunk_plan& B, WJCP* out) {
  if (int rc = ecsimd_hip_memcpy_d2h(S.c, static_cast<void*>(out), B.dow, B.pb)) die("d2h", rc, S.c);   // (waits for this side's ladder, then copies)
}
void batch_raw(std::span<const WBN> x, std::span<const WJCP> P, std::span<WJCP> out) {
  const size_t wides = x.size(), rec = layout().rec;
  if (wides <= CHUNK_WIDES) {
    side& S = side_of(0);
    const chunk_plan B = plan(S, wides, rec);
    launch_chunk(S, B, x.data(), P.data(), wides);
    drain_chunk(S, B, out.data());
    return;
  }
  const size_t chunks = (wides + CHUNK_WIDES - 1) / CHUNK_WIDES;
  chunk_plan B[2];
  const int option = ecsimd_hip_get_ref_square_compat(side_of(0).c);          // whatever the caller set on the adapter's context holds for the second one too
  for (int i = 0; i < 2; ++i) { side& S = side_of(i); if (int rc = ecsimd_hip_set_ref_square_compat(S.c, option)) die("set_ref_square_compat", rc, S.c); B[i] = plan(S, CHUNK_WIDES, rec); }
  for (size_t c = 0; c <= chunks; ++c) {
    if (c < chunks) {
      const size_t first = c * CHUNK_WIDES, m = (wides - first) < CHUNK_WIDES ? (wides - first) : CHUNK_WIDES;
      chunk_plan b = B[c & 1]; b.pb = m * rec;
      launch_chunk(side_of((int)(c & 1)), b, x.data() + first, P.data() + first, m);
    }
    if (c > 0) {
      const size_t first = (c - 1) * CHUNK_WIDES, m = (wides - first) < CHUNK_WIDES ? (wides - first) : CHUNK_WIDES;
      chunk_plan b = B[(c - 1) & 1]; b.pb = m * rec;
      drain_chunk(side_of((int)((c - 1) & 1)), b, out.data() + first);
    }
  }
}
}  // namespace

ecsimd_hip_ctx* scalar_mult_p256_context() { return ctx(); }
void scalar_mult_p256_set_ref_square_compat(bool on) {
  if (int rc = ecsimd_hip_set_ref_square_compat(ctx(), on ? 1 : 0)) die("set_ref_square_compat", rc, ctx());
}
bool scalar_mult_p256_transposes_on_the_device() { return layout().ok; }

void scalar_mult_p256(std::span<const WBN> x, std::span<const WJCP> P, std::span<WJCP> out) {
  const size_t wides = x.size(), n = 4 * wides;
  if (P.size() != wides || out.size() != wides) die("span lengths", -1, nullptr);
  if (!n) return;
  if (layout().ok) { batch_raw(x, P, out); return; }
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
