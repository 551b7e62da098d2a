// adapter_driver.cpp -- runs integration/scalar_mult_p256_adapter.cpp on a GPU, beside the REAL reference in the same process.
//
// TEST INFRASTRUCTURE (oracle/README.md): built only where /root/reference exists (oracle/Makefile, output oracle/_ref/adapter_driver, which
// travels to the GPU box as a built artefact like oracle/_ref/libecsimd_ref.so).  This translation unit includes the reference's headers
// where they lie and contains none of its code.  tests/test_integration_adapter.py runs the binary (-m gpu).
//   1. the reference's own ScalarMult scenarios (tests/curve_group.cpp:117-173): scalar_mult_p256(x, WJG) -> to_affine() against the
//      expected affine points the reference's test holds (the same hex vectors as tests/golden/reference_kats.json);
//   2. `wides` lane-distinct wides (scalars and base points different in every lane) through the adapter's batch form AND its four-lane form,
//      X, Y, Z compared limb for limb with curve_group<curve_nist_p256>::scalar_mult computed by the reference in this process --
//      with ECSIMD_HIP_REF_SQUARE_COMPAT set through the C ABI, so that not one limb may differ (the reference's square() drops a carry on
//      ~3e-6 of random scalar multiplications: DESIGN.md section 5), then once more without it (exact squaring; differing lanes are reported);
//   3. the batch form's rate, staging and PCIe included: `rate_wides` wides in one call (argv[2]; argv[3]: a second size).  ECSIMD_ADAPTER_HOST_TRANSPOSE=1
//      selects the adapter's per-lane host conversion instead of the device transposition (the A/B).
#include "../integration/scalar_mult_p256_adapter.h"

#include <ecsimd/literals.h>
#include <ecsimd/serialization.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ecsimd;
using namespace ecsimd::literals;
using namespace ecsimd_mi355x;
using CG = curve_group<Curve>;
using BN = typename WBN::value_type;
using WCP = wide_curve_point<Curve>;

namespace {
int failures = 0;
#define CHECK(cond) do { if (!(cond)) { ++failures; std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } } while (0)

uint64_t splitmix64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
BN random_bn(uint64_t stream, uint64_t index) {
  typename BN::cbn_type c;
  for (int l = 0; l < 4; ++l) c[l] = splitmix64(0x5EEDEC51D0000001ull ^ (stream << 56) ^ (4 * index + l));
  return BN::from(c);
}
WBN random_wide(uint64_t stream, uint64_t w) { return WBN{[&](auto lane, auto) { return random_bn(stream, 4 * w + lane); }}; }
bool same(WBN const& a, WBN const& b) { return eve::all(a == b); }
size_t lanes_differing(WJCP const& a, WJCP const& b) {
  size_t d = 0;
  for (int lane = 0; lane < 4; ++lane) {
    bool eq = true;
    for (auto pr : {std::pair{&a.x(), &b.x()}, std::pair{&a.y(), &b.y()}, std::pair{&a.z(), &b.z()}}) {
      const auto u = pr.first->wbn().get(lane).cbn(), v = pr.second->wbn().get(lane).cbn();
      for (int l = 0; l < 4; ++l) eq = eq && u[l] == v[l];
    }
    d += !eq;
  }
  return d;
}
}  // namespace

int main(int argc, char** argv) {
  const size_t wides = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 256;
  const size_t rate_wides0 = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : (size_t)1 << 16;

  // ---- 1. tests/curve_group.cpp:117-173 (ScalarMult) through the adapter
  {
    const auto WJG = CG::WJG();
    struct kat { BN k, x, y; };
    const kat kats[] = {
      {bn_from_bytes_BE<BN>("0000000000000000000000000000000000000000000000000000000000000005"_hex),
       bn_from_bytes_BE<BN>("51590b7a515140d2d784c85608668fdfef8c82fd1f5be52421554a0dc3d033ed"_hex), bn_from_bytes_BE<BN>("e0c17da8904a727d8ae1bf36bf8a79260d012f00d4d80888d1d0bb44fda16da4"_hex)},
      {bn_from_bytes_BE<BN>("0bc1b1f28709decb543d9677d2cc9942348f6b984deff409430740942ff38827"_hex),
       bn_from_bytes_BE<BN>("1b7721565b2c4a9f203bbccc6b531df2789fde0d135c76db71e4a7bbab9e85b2"_hex), bn_from_bytes_BE<BN>("393655bcc30f67f3a4e257b39685657d7c8df7b2a132b49c848003e300c8dcd1"_hex)},
      {bn_from_bytes_BE<BN>("0a891cecc2bf13b0aca744434a9c9f4bd7bf5c8ed86e2f76e7df72bad813bd80"_hex),
       bn_from_bytes_BE<BN>("f411d79e2997b2954975046d23b0e4a69ce580a4a81e1bed18fef6fd9ea4a912"_hex), bn_from_bytes_BE<BN>("43895f527937e816c3d7c0a2370002796d3cd4860cb034df86cbe7da227d9113"_hex)},
    };
    for (auto const& t : kats) {
      const auto WJP = scalar_mult_p256(WBN{t.k}, WJG);                 // the adapter; Jacobian, Montgomery form
      const auto WP = WJP.to_affine();                                  // the REFERENCE's to_affine on the adapter's result
      CHECK(same(WP.x(), WBN{t.x})); CHECK(same(WP.y(), WBN{t.y}));
      CHECK(lanes_differing(WJP, CG::scalar_mult(WBN{t.k}, WJG)) == 0);  // and the Jacobian coordinates are the reference's own, limb for limb
    }
    std::printf("scenarios of tests/curve_group.cpp ScalarMult through the adapter: %s\n", failures ? "FAILED" : "ok");
  }

  // ---- 2. lane-distinct wides against the reference in this process
  std::vector<WBN> x(wides); std::vector<WJCP> P(wides), got(wides), ref(wides);
  {
    const auto WJG = CG::WJG();
    for (size_t w = 0; w < wides; ++w) {
      x[w] = random_wide(1, w);
      P[w] = WJCP::from_affine(CG::scalar_mult(random_wide(2, w), WJG).to_affine());     // P_i = s_i * G, a different point in every lane (reference arithmetic)
      ref[w] = CG::scalar_mult(x[w], P[w]);
    }
    // a few hand-picked lanes on top: k = 1, 2, an even k, n - 2, 2^256 - 1
    if (wides) x[0] = WBN{[&](auto lane, auto) { const char* h[4] = {"0000000000000000000000000000000000000000000000000000000000000001", "0000000000000000000000000000000000000000000000000000000000000002",
        "ffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc63254f", "ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff"};
        typename BN::cbn_type c; for (int l = 0; l < 4; ++l) { c[l] = 0; for (int j = 0; j < 16; ++j) { const char ch = h[lane][(3 - l) * 16 + j]; c[l] = (c[l] << 4) | (uint64_t)(ch <= '9' ? ch - '0' : ch - 'a' + 10); } }
        return BN::from(c); }};
    if (wides) ref[0] = CG::scalar_mult(x[0], P[0]);
  }
  CHECK(scalar_mult_p256_context() != nullptr);
  size_t diff_compat = 0, diff_exact = 0, diff_wide = 0;
  scalar_mult_p256_set_ref_square_compat(true);
  scalar_mult_p256(std::span<const WBN>(x), std::span<const WJCP>(P), std::span<WJCP>(got));            // the batch form: ONE launch
  for (size_t w = 0; w < wides; ++w) diff_compat += lanes_differing(got[w], ref[w]);
  for (size_t w = 0; w < wides && w < 8; ++w) diff_wide += lanes_differing(scalar_mult_p256(x[w], P[w]), ref[w]);     // the reference's own signature
  CHECK(diff_compat == 0); CHECK(diff_wide == 0);
  scalar_mult_p256_set_ref_square_compat(false);
  scalar_mult_p256(std::span<const WBN>(x), std::span<const WJCP>(P), std::span<WJCP>(got));
  for (size_t w = 0; w < wides; ++w) diff_exact += lanes_differing(got[w], ref[w]);
  std::printf("lane-distinct: %zu lanes through the batch form; differing from the in-process reference: %zu with REF_SQUARE_COMPAT (must be 0), %zu with exact squaring "
              "(the reference's dropped carry: ~3e-6 per lane); four-lane form on %zu wides: %zu differing\n", 4 * wides, diff_compat, diff_exact, wides < 8 ? wides : (size_t)8, diff_wide);
  CHECK(diff_exact <= 1 + wides / 1000);

  // ---- 3. the batch form's rate (AoSoA <-> AoS conversion, H2D, one launch, D2H: everything a caller pays); argv[3]: a second, larger size
  std::printf("lane transposition: %s\n", scalar_mult_p256_transposes_on_the_device() ? "on the device (the spans travel as raw bytes: the layout check passed)" : "per lane on the host");
  if (!std::getenv("ECSIMD_ADAPTER_HOST_TRANSPOSE")) CHECK(scalar_mult_p256_transposes_on_the_device());
  for (int arg = 2; arg < 4; ++arg) {
    const size_t rate_wides = arg == 2 ? rate_wides0 : (argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 0);
    if (!rate_wides) continue;
    std::vector<WBN> bx(rate_wides); std::vector<WJCP> bP(rate_wides, P.empty() ? CG::WJG() : P[0]), bout(rate_wides);
    for (size_t w = 0; w < rate_wides; ++w) { bx[w] = random_wide(3, w); if (!P.empty()) bP[w] = P[w % wides]; }
    scalar_mult_p256(std::span<const WBN>(bx), std::span<const WJCP>(bP), std::span<WJCP>(bout));       // warm-up (sizes the staging)
    const auto t0 = std::chrono::steady_clock::now();
    scalar_mult_p256(std::span<const WBN>(bx), std::span<const WJCP>(bP), std::span<WJCP>(bout));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const auto r0 = std::chrono::steady_clock::now();
    size_t sample = rate_wides < 64 ? rate_wides : 64, bad = 0;
    auto pick = [&](size_t i) { return i + 1 == sample ? rate_wides - 1 : i * (rate_wides / sample) + (i % 3); };      // spread over the batch (every chunk of a large one), the last wide included
    for (size_t i = 0; i < sample; ++i) { const size_t w = pick(i); bad += lanes_differing(bout[w], CG::scalar_mult(bx[w], bP[w])) > 1; }       // (a sample of the big batch against the reference as well)
    const double rdt = std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
    CHECK(bad == 0);
    std::printf("batch form: %zu wides = %zu lanes in one call: %.3f ms = %.2f M scalar mults/s (conversion + PCIe + launch); the reference on this host thread: %.2f k/s\n",
                rate_wides, 4 * rate_wides, 1e3 * dt, 4e-6 * rate_wides / dt, 4e-3 * sample / rdt);
    if (arg == 3) {                                                     // the large batch once more with the reference's squaring: the sampled wides may not differ in one limb
      scalar_mult_p256_set_ref_square_compat(true);
      scalar_mult_p256(std::span<const WBN>(bx), std::span<const WJCP>(bP), std::span<WJCP>(bout));
      scalar_mult_p256_set_ref_square_compat(false);
      size_t d = 0;
      for (size_t i = 0; i < sample; ++i) { const size_t w = pick(i); d += lanes_differing(bout[w], CG::scalar_mult(bx[w], bP[w])); }
      CHECK(d == 0);
      std::printf("the same batch with REF_SQUARE_COMPAT: %zu sampled wides spread over the batch, %zu lanes differ (must be 0)\n", sample, d);
    }
  }
  std::printf(failures ? "adapter_driver: %d check(s) FAILED\n" : "adapter_driver ok (%d failed)\n", failures);
  return failures ? 1 : 0;
}
