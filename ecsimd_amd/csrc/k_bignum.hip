// k_bignum.hip -- curve-independent 256-bit kernels (reference layer L2: add.h, sub.h, shift.h,
// mul.h, swap.h), the synthetic-input generator and the integer-multiply peak probe.
#include "kernels.h"
#include "field.cuh"

namespace ecsimd_hip {
namespace {
using launch::BLOCK;
#define GID size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= n) return

__global__ void __launch_bounds__(BLOCK) k_add(const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* carry, size_t n) {
  GID; fe x = fe_load(a, i); const fe y = fe_load(b, i);
  const uint32_t c = add8(x, y); fe_store(out, i, x); if (carry) carry[i] = (uint8_t)c;
}
__global__ void __launch_bounds__(BLOCK) k_sub(const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* borrow, size_t n) {
  GID; fe x = fe_load(a, i); const fe y = fe_load(b, i);
  const uint32_t m = sub8(x, y); if (out) fe_store(out, i, x); if (borrow) borrow[i] = (uint8_t)(m & 1u);
}
__global__ void __launch_bounds__(BLOCK) k_sub_if_above(const uint64_t* a, const uint64_t* p, uint64_t* out, size_t n) {
  GID; const fe x = fe_load(a, i); const fe y = fe_load(p, i);
  fe d = x; const uint32_t m = sub8(d, y);          // m = all-ones when a < p: keep a   (sub.h:46-69)
  fe_store(out, i, fe_select(m, x, d));
}
__global__ void __launch_bounds__(BLOCK) k_shift_left_one(const uint64_t* a, uint64_t* out, uint8_t* carry, size_t n) {
  GID; const fe x = fe_load(a, i); fe s;
#pragma unroll
  for (int j = 7; j > 0; --j) s.w[j] = __builtin_amdgcn_alignbit(x.w[j], x.w[j - 1], 31);
  s.w[0] = x.w[0] << 1;
  fe_store(out, i, s); if (carry) carry[i] = (uint8_t)(x.w[7] >> 31);
}
__global__ void __launch_bounds__(BLOCK) k_mul(const uint64_t* a, const uint64_t* b, uint64_t* out8, size_t n) {
  GID; fe2_store(out8, i, mul8x8(fe_load(a, i), fe_load(b, i)));
}
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_square(const uint64_t* a, uint64_t* out8, size_t n) {
  GID; if constexpr (REF) fe2_store(out8, i, sqr8_ref(fe_load(a, i))); else fe2_store(out8, i, sqr8(fe_load(a, i)));
}
__global__ void __launch_bounds__(BLOCK) k_swap_if(const uint8_t* mask, uint64_t* a, uint64_t* b, size_t n) {
  GID; fe x = fe_load(a, i), y = fe_load(b, i);
  fe_cswap(0u - (uint32_t)(mask[i] != 0), x, y); fe_store(a, i, x); fe_store(b, i, y);
}

// lane-wise equality of `limbs`-limb elements (4 or 8), lane-mask logic and the count behind all() / any():
// the reference gets these from eve's logical<wide> (bignum.h:136-137, eve::all / eve::any in its tests).
__global__ void __launch_bounds__(BLOCK) k_cmp_eq(const uint64_t* a, const uint64_t* b, int limbs, uint8_t* flag, size_t n) {
  GID; const int stride = limbs <= 4 ? 4 : 8;          // narrower values (the 128- and 192-bit types of the reference's unit tests)
  const uint64_t* x = a + (size_t)stride * i;          // travel in the low limbs of a 256-bit element
  const uint64_t* y = b + (size_t)stride * i;
  uint64_t d = 0;
  for (int j = 0; j < limbs; ++j) d |= x[j] ^ y[j];
  flag[i] = (uint8_t)(d == 0ull);
}
__global__ void __launch_bounds__(BLOCK) k_mask_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  GID; const bool x = a[i] != 0, y = b ? b[i] != 0 : false;
  out[i] = (uint8_t)(op == 0 ? !x : op == 1 ? (x && y) : op == 2 ? (x || y) : (x == y));
}
__global__ void __launch_bounds__(BLOCK) k_mask_count(const uint8_t* a, size_t n, unsigned long long* count) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool set = i < n && a[i] != 0;
  const unsigned long long lanes = __builtin_amdgcn_ballot_w64(set);
  if ((threadIdx.x & 63) == 0 && lanes) atomicAdd(count, (unsigned long long)__builtin_popcountll(lanes));
}

// word w of element i of stream s = splitmix64(seed ^ (s << 56) ^ (4 i + w))
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(BLOCK) k_fill_random(uint64_t* out, size_t n, uint64_t seed, uint64_t stream, uint64_t first, int clear_top) {
  GID;
  uint64_t v[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) v[w] = splitmix64(seed ^ (stream << 56) ^ ((first + i) * 4 + (uint64_t)w));
  if (clear_top > 0) v[3] &= (~0ull) >> clear_top;
  uint4* p = reinterpret_cast<uint4*>(out + 4 * i);
  p[0] = make_uint4((uint32_t)v[0], (uint32_t)(v[0] >> 32), (uint32_t)v[1], (uint32_t)(v[1] >> 32));
  p[1] = make_uint4((uint32_t)v[2], (uint32_t)(v[2] >> 32), (uint32_t)v[3], (uint32_t)(v[3] >> 32));
}

// 16 independent 64-bit accumulators per lane, carries never consumed: the pure v_mad_u64_u32
// issue rate, i.e. the denominator of the integer-multiply roofline (SURVEY.md 8(d) "Peak").
__global__ void __launch_bounds__(BLOCK) k_peak_mad32(uint32_t* sink, int iters, uint32_t seed) {
  uint64_t a[16];
  const uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = (uint64_t)(b * (j + 1) + c) * 0x100000001ull;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c) : "vcc");
  }
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s ^= a[j];
  if (s == 0x12345678u) sink[threadIdx.x] = (uint32_t)s;   // practically never true; keeps the chains live
}
}  // namespace

__global__ void __launch_bounds__(BLOCK) k_if_else(const uint8_t* mask, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  GID; fe_store(out, i, fe_select(0u - (uint32_t)(mask[i] != 0), fe_load(a, i), fe_load(b, i)));      // ifelse.h:15-22
}

// The small-batch route of scalar_mult_base (capi.hip): the comb's affine result is the true k*G, the reference ladder's differs from it at its
// three degenerate scalars (curve_group.h:189-218 as written).  `special` = 3 scalars, then the ladder's affine x and y for each (9 elements):
// a lane whose k equals one of the scalars takes the ladder's coordinates -- by selects, no branch on k.
__global__ void __launch_bounds__(BLOCK) k_patch_special(const uint64_t* k, const uint64_t* special, uint64_t* ox, uint64_t* oy, size_t n) {
  GID; const fe v = fe_load(k, i);
  fe x = fe_load(ox, i), y;
  if (oy) y = fe_load(oy, i);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const uint32_t m = 0u - (uint32_t)fe_eq(v, fe_load(special, j));
    x = fe_select(m, fe_load(special, 3 + j), x);
    if (oy) y = fe_select(m, fe_load(special, 6 + j), y);
  }
  fe_store(ox, i, x);
  if (oy) fe_store(oy, i, y);
}

namespace launch {
static_assert(PEAK_MADS_PER_LANE_PER_ITER == 4 * 16, "keep in sync with k_peak_mad32");
#define GO(kern, ...) hipLaunchKernelGGL(kern, grid_for(n), dim3(BLOCK), 0, s, __VA_ARGS__)
void add(hipStream_t s, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* carry, size_t n) { GO(k_add, a, b, out, carry, n); }
void sub(hipStream_t s, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* borrow, size_t n) { GO(k_sub, a, b, out, borrow, n); }
void sub_if_above(hipStream_t s, const uint64_t* a, const uint64_t* p, uint64_t* out, size_t n) { GO(k_sub_if_above, a, p, out, n); }
void shift_left_one(hipStream_t s, const uint64_t* a, uint64_t* out, uint8_t* carry, size_t n) { GO(k_shift_left_one, a, out, carry, n); }
void mul(hipStream_t s, const uint64_t* a, const uint64_t* b, uint64_t* out8, size_t n) { GO(k_mul, a, b, out8, n); }
void square(hipStream_t s, const uint64_t* a, uint64_t* out8, size_t n, bool ref_compat) { if (ref_compat) GO(k_square<true>, a, out8, n); else GO(k_square<false>, a, out8, n); }
void swap_if(hipStream_t s, const uint8_t* mask, uint64_t* a, uint64_t* b, size_t n) { GO(k_swap_if, mask, a, b, n); }
void if_else(hipStream_t s, const uint8_t* mask, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { GO(k_if_else, mask, a, b, out, n); }
void patch_special(hipStream_t s, const uint64_t* k, const uint64_t* special, uint64_t* ox, uint64_t* oy, size_t n) { GO(k_patch_special, k, special, ox, oy, n); }
void cmp_eq(hipStream_t s, const uint64_t* a, const uint64_t* b, int limbs, uint8_t* flag, size_t n) { GO(k_cmp_eq, a, b, limbs, flag, n); }
void mask_op(hipStream_t s, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) { GO(k_mask_op, op, a, b, out, n); }
void mask_count(hipStream_t s, const uint8_t* a, size_t n, unsigned long long* count) { GO(k_mask_count, a, n, count); }
void fill_random(hipStream_t s, uint64_t* out, size_t n, uint64_t seed, uint64_t stream, uint64_t first, int clear_top) { GO(k_fill_random, out, n, seed, stream, first, clear_top); }
void peak_mad32(hipStream_t s, int blocks, uint32_t* sink, int iters, uint32_t seed) {
  hipLaunchKernelGGL(k_peak_mad32, dim3(blocks), dim3(BLOCK), 0, s, sink, iters, seed);
}
}  // namespace launch
}  // namespace ecsimd_hip
