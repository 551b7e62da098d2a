// k_gladder.hip -- the scalar-multiplication ladder for a curve registered at RUN time (curve_group.h:189-251 instantiated with any Curve type).
//
// One kernel serves scalar_mult, scalar_mult_1s (k_stride = 0) and scalar_mult_base (x == nullptr: the curve's generator), as k_ladder.inc does
// for the built-in curves.  RADIX 29 (default): the 254 ZDAU iterations on fe29.cuh's nine signed 29-bit limbs with the dense prime in SGPRs;
// RADIX 32: on gfield.cuh's canonical words (ECSIMD_HIP_LADDER_RADIX32: the A/B); <32, true>: with the reference's square() as written
// (ECSIMD_HIP_REF_SQUARE_COMPAT: the reference's bits on every lane).  Same X, Y, Z from all three wherever square(a) == mul(a, a).
#include "kernels.h"
#include "gcurve.cuh"
#include "../../include/ecsimd_hip.h"

namespace ecsimd_hip {
namespace {
using launch::BLOCK;
#ifndef GLADDER_WAVES_PER_SIMD
#define GLADDER_WAVES_PER_SIMD 3
#endif

// The canonical-word loops want 188 / 178 registers: at 3 waves per SIMD 27 / 11 of them spill (constant-addressed scratch, re-read in every iteration: 6.7 /
// 4.7 GB per 2^24 launch where 3.2 are algorithmic) -- and are still no slower than spill-free at 2 waves (20.3 against 19.8 M/s with the reference's squaring,
// 20.66 against 20.63 without: profiles/r05/ab_generic_radix32_ladders_waves_per_simd.txt).  These are the A/B and bit-identical forms; the default loop spills nothing.
#ifndef GLADDER32_WAVES_PER_SIMD
#define GLADDER32_WAVES_PER_SIMD 3
#endif
template <int RADIX, bool REF> __global__ void __launch_bounds__(BLOCK, RADIX == 29 ? GLADDER_WAVES_PER_SIMD : GLADDER32_WAVES_PER_SIMD)
k_gc_scalar_mult(gcurve G, const uint64_t* __restrict__ k, int k_stride, const uint64_t* __restrict__ x, const uint64_t* __restrict__ y,
                 uint64_t* __restrict__ ox, uint64_t* __restrict__ oy, uint64_t* __restrict__ oz, size_t n, int flags) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  fe xm, ym;
  if (x == nullptr) { xm = g_from_classical(g_words(G.gx), G.F); ym = g_from_classical(g_words(G.gy), G.F); }
  else {
    xm = fe_load(x, i); ym = fe_load(y, i);
    if (!(flags & ECSIMD_HIP_BASE_MGRY)) { xm = g_from_classical(xm, G.F); ym = g_from_classical(ym, G.F); }
  }
  const uint32_t* kw = reinterpret_cast<const uint32_t*>(k + (size_t)k_stride * i);
  const gjpoint R = gc_scalar_mult_ladder<RADIX, REF>(kw, xm, ym, G);
  fe_store(ox, i, R.x); fe_store(oy, i, R.y); fe_store(oz, i, R.z);
}

// The loop body alone, `iters` times in registers (ecsimd_hip_zdau_repeat on a registered curve): the ladder's loop with a public swap pattern.
template <int RADIX> __global__ void __launch_bounds__(BLOCK, GLADDER_WAVES_PER_SIMD)
k_gc_zdau_repeat(gcurve G, const uint64_t* __restrict__ px, const uint64_t* __restrict__ py, const uint64_t* __restrict__ pz, const uint64_t* __restrict__ qx, const uint64_t* __restrict__ qy,
                 uint64_t* __restrict__ rx, uint64_t* __restrict__ ry, uint64_t* __restrict__ sx, uint64_t* __restrict__ sy, uint64_t* __restrict__ oz, size_t n, int iters, uint64_t swap_bits) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  fe x1 = fe_load(px, i), y1 = fe_load(py, i), z = fe_load(pz, i), x2 = fe_load(qx, i), y2 = fe_load(qy, i);
  if constexpr (RADIX == 29) {
    constexpr int C = CURVE_GENERIC;
    const r29_ctx<C>& cx = G.r29;
    coz29 s;
    s.x1 = enter29<C>(x1, cx); s.x2 = enter29<C>(x2, cx); s.y1 = enter29<C>(y1, cx); s.z = enter29<C>(z, cx);
    s.dx = sub29(s.x1, s.x2);
    s.dy = sub29(s.y1, enter29<C>(y2, cx));
#pragma unroll 1
    for (int t = 0; t < iters; ++t) zdau29<C>(s, 0u - (uint32_t)((swap_bits >> (t & 63)) & 1u), cx);
    x1 = leave29<C>(s.x1, cx); y1 = leave29<C>(s.y1, cx); x2 = leave29<C>(s.x2, cx); y2 = leave29<C>(sub29(s.y1, s.dy), cx); z = leave29<C>(s.z, cx);
  } else {
#pragma unroll 1
    for (int t = 0; t < iters; ++t) gc_zdau<false>(x1, y1, x2, y2, z, G, 0u - (uint32_t)((swap_bits >> (t & 63)) & 1u));
  }
  fe_store(rx, i, x1); fe_store(ry, i, y1); fe_store(sx, i, x2); fe_store(sy, i, y2); fe_store(oz, i, z);
}
}  // namespace

namespace launch {
// flags: ECSIMD_HIP_BASE_MGRY | ECSIMD_HIP_LADDER_RADIX32 | ECSIMD_HIP_REF_SQUARE_COMPAT; Jacobian Montgomery out
void gc_scalar_mult(hipStream_t s, const gcurve& G, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  if (flags & ECSIMD_HIP_REF_SQUARE_COMPAT) hipLaunchKernelGGL((k_gc_scalar_mult<32, true>), grid_for(n), dim3(BLOCK), 0, s, G, k, k_stride, x, y, ox, oy, oz, n, flags);
  else if (flags & ECSIMD_HIP_LADDER_RADIX32) hipLaunchKernelGGL((k_gc_scalar_mult<32, false>), grid_for(n), dim3(BLOCK), 0, s, G, k, k_stride, x, y, ox, oy, oz, n, flags);
  else hipLaunchKernelGGL((k_gc_scalar_mult<29, false>), grid_for(n), dim3(BLOCK), 0, s, G, k, k_stride, x, y, ox, oy, oz, n, flags);
}
void gc_zdau_repeat(hipStream_t s, const gcurve& G, const uint64_t* px, const uint64_t* py, const uint64_t* pz, const uint64_t* qx, const uint64_t* qy,
                    uint64_t* rx, uint64_t* ry, uint64_t* sx, uint64_t* sy, uint64_t* oz, size_t n, int iters, uint64_t swap_bits, int radix) {
  if (radix == 29) hipLaunchKernelGGL(k_gc_zdau_repeat<29>, grid_for(n), dim3(BLOCK), 0, s, G, px, py, pz, qx, qy, rx, ry, sx, sy, oz, n, iters, swap_bits);
  else hipLaunchKernelGGL(k_gc_zdau_repeat<32>, grid_for(n), dim3(BLOCK), 0, s, G, px, py, pz, qx, qy, rx, ry, sx, sy, oz, n, iters, swap_bits);
}
}  // namespace launch
}  // namespace ecsimd_hip
