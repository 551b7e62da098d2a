// benchs/host_benchs.cpp -- the reference's two benchmark programs re-hosted on this repository's C++ host API
// (include/ecsimd/*.h over the C ABI): what benchs/curve_group.cpp:23-48 and benchs/ops.cpp:36-100 time, with the
// same calls, at a runtime batch size instead of the reference's four lanes.  No Google Benchmark here: a plain
// wall-clock loop fenced by ecsimd::hip::sync().  Includes host-side allocation of the results, like the reference's
// by-value returns.
//
//   g++ -std=c++20 -O2 -I include benchs/host_benchs.cpp -L ecsimd_amd -lecsimd_hip -Wl,-rpath,$PWD/ecsimd_amd -o build/host_benchs
//   build/host_benchs [log2_batch=20] [repetitions=5]
#include <ecsimd/ecsimd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

using namespace ecsimd;
using namespace ecsimd::literals;
using W256 = wide_bignum<bignum_256>;
using W512 = wide_bignum<bignum_512>;

namespace {
template <class F> double seconds_per_call(F&& body, int reps) {
  body(); hip::sync();                                             // warm-up: tables, workspace
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) body();
  hip::sync();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
}
void report(const char* name, size_t n, double s, const char* unit) {
  std::printf("%-34s batch %9zu  %10.3f ms/call  %12.1f M %s/s\n", name, n, 1e3 * s, n / s / 1e6, unit);
}
uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
W256 random_batch(size_t n, uint64_t stream, bool clear_top_byte) {  // benchs/ops.cpp random_bn<BN, LastZero>
  return W256(n, [=](size_t i, size_t) {
    bignum_256 b;
    for (int l = 0; l < 4; ++l) b.limbs[l] = mix(stream * 0x100000000ull + 4 * i + l);
    if (clear_top_byte) b.limbs[3] &= 0x00ffffffffffffffull;
    return b;
  });
}
}  // namespace

int main(int argc, char** argv) {
  const int lg = argc > 1 ? std::atoi(argv[1]) : 20;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
  if (lg < 2 || lg > 24 || reps < 1) { std::fprintf(stderr, "usage: %s [log2_batch 2..24] [repetitions]\n", argv[0]); return 2; }
  const size_t n = (size_t)1 << lg;
  using Curve = curve_nist_p256; using CG = curve_group<Curve>;

  // ---- benchs/curve_group.cpp: scalar_mult(x, WJG).to_affine() and the one-scalar variant
  const auto k1 = bn_from_bytes_BE<bignum_256>("0a891cecc2bf13b0aca744434a9c9f4bd7bf5c8ed86e2f76e7df72bad813bd80"_hex);
  const auto G = CG::WJG(n);
  const W256 x(n, k1);
  report("scalar_mult_p256 + to_affine", n, seconds_per_call([&] { (void)CG::scalar_mult(x, G).to_affine(); }, reps), "scalar mults");
  report("scalar_mult_p256_1s + to_affine", n, seconds_per_call([&] { (void)CG::scalar_mult_1s(k1, G).to_affine(); }, reps), "scalar mults");
  const auto ks = random_batch(n, 1, false);
  const auto P = CG::scalar_mult_base_affine(random_batch(n, 2, false));           // lane-distinct points
  report("  (ext) scalar_mult_affine, windowed", n, seconds_per_call([&] { (void)CG::scalar_mult_affine(ks, P); }, reps), "scalar mults");
  report("  (ext) scalar_mult_base_affine", n, seconds_per_call([&] { (void)CG::scalar_mult_base_affine(ks); }, reps), "scalar mults");

  // ---- benchs/ops.cpp: add_256, mul_256, sqr_256, mgry_sqr_256, mgry_reduce_512 over the secp256k1 prime (:22-24)
  using K = curve_secp256k1;
  const auto a = random_batch(n, 3, true), b = random_batch(n, 4, true);
  report("add_256", n, seconds_per_call([&] { (void)add(a, b); }, reps), "elements");
  report("mul_256", n, seconds_per_call([&] { (void)mul(a, b); }, reps), "elements");
  report("sqr_256", n, seconds_per_call([&] { (void)square(a); }, reps), "elements");
  const wide_mgry_bignum<W256, K::P> am{a};
  report("mgry_sqr_256", n, seconds_per_call([&] { (void)mgry_sqr(am); }, reps), "elements");
  const W512 wide_product = mul(a, b);
  report("mgry_reduce_512", n, seconds_per_call([&] { (void)details::mgry_reduce<K::P>(wide_product); }, reps), "elements");
  return 0;
}
