// k_serial.hip -- wire formats on the device (reference host code:
// serialization.h:12-48 bn_from_bytes_BE / bn_to_bytes_BE, utility.h:45-51 wide_mask_bit).
//
//  * k_bytes_be: 32 big-endian bytes <-> 4 x u64 little-endian limbs.  The map is an involution
//    (reverse the eight 32-bit words, byte-swap each), so one kernel serves both directions.
//    HBM-bound: 32 B in + 32 B out per element, two 16-byte accesses per lane each way.
//  * SEC1 v2 2.3.3 / 2.3.4 elliptic-curve points: 04 || X || Y (65 B) and 02/03 || X (33 B).  The
//    records are not a multiple of 4 bytes, so a workgroup moves its 256 records through LDS: coalesced
//    dword traffic on the HBM side, byte accesses on the LDS side.  Decoding validates (x, y < p, on the
//    curve) or decompresses (y = sqrt(x^3 + a x + b), parity from the prefix) with a per-lane ok flag.
#include "kernels.h"
#include "point.cuh"
#include "gcurve.cuh"

namespace ecsimd_hip {
namespace {
using launch::BLOCK;
#define GID size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= n) return

__global__ void __launch_bounds__(BLOCK) k_bytes_be(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  GID;
  const uint4 a = in[2 * i], b = in[2 * i + 1];                    // words 0..3, 4..7 of the record
  out[2 * i]     = make_uint4(__builtin_bswap32(b.w), __builtin_bswap32(b.z), __builtin_bswap32(b.y), __builtin_bswap32(b.x));
  out[2 * i + 1] = make_uint4(__builtin_bswap32(a.w), __builtin_bswap32(a.z), __builtin_bswap32(a.y), __builtin_bswap32(a.x));
}

__global__ void __launch_bounds__(BLOCK) k_mask_bit(const uint64_t* __restrict__ a, int bit, uint8_t* __restrict__ flag, size_t n) {
  GID;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(a + 4 * i);
  flag[i] = (uint8_t)((w[bit >> 5] >> (bit & 31)) & 1u);
}

// ---- the reference's register layout <-> one element per lane (round 5) -----------------------
// The reference's wide_bignum<bignum_256> is eve::wide<bignum, fixed<4>>: four 256-bit lanes stored limb-major, u64[limb * 4 + lane]
// (bignum.h:99-100, eve/arch/cpu/as_register.hpp:55-60) -- 128 bytes per wide, and a Jacobian point is three of them.  A caller that keeps
// reference types copies its array of wides to the device AS IT IS and these two kernels do the 4 x 4 transposition at HBM speed: wide w of a
// record array (record_bytes apart, the wide at offset_bytes) <-> elements 4w .. 4w + 3 of the ABI's array (u64[4 * e + limb]).
// One thread per ELEMENT: its 32 bytes of the ABI side are two 16-byte accesses, consecutive threads consecutive; on the wide side the four
// threads of a wide cover 32 contiguous bytes per limb.
__global__ void __launch_bounds__(BLOCK) k_wide4_to_lanes(const uint8_t* __restrict__ wides, size_t record_bytes, size_t offset_bytes, uint4* __restrict__ out, size_t n) {
  GID;
  const uint2* w = reinterpret_cast<const uint2*>(wides + (i >> 2) * record_bytes + offset_bytes) + (i & 3);
  const uint2 l0 = w[0], l1 = w[4], l2 = w[8], l3 = w[12];
  out[2 * i] = make_uint4(l0.x, l0.y, l1.x, l1.y);
  out[2 * i + 1] = make_uint4(l2.x, l2.y, l3.x, l3.y);
}
__global__ void __launch_bounds__(BLOCK) k_lanes_to_wide4(const uint4* __restrict__ in, uint8_t* __restrict__ wides, size_t record_bytes, size_t offset_bytes, size_t n) {
  GID;
  const uint4 a = in[2 * i], b = in[2 * i + 1];
  uint2* w = reinterpret_cast<uint2*>(wides + (i >> 2) * record_bytes + offset_bytes) + (i & 3);
  w[0] = make_uint2(a.x, a.y); w[4] = make_uint2(a.z, a.w); w[8] = make_uint2(b.x, b.y); w[12] = make_uint2(b.z, b.w);
}

// ---- SEC1 records through LDS -------------------------------------------------------------
template <int REC> struct rec_lds {            // 256 records of REC bytes, dword-granular staging
  static constexpr int BYTES = BLOCK * REC;    // 8448 (33) or 16640 (65): both multiples of 4
  static constexpr int DWORDS = BYTES / 4;
};
template <int REC> __device__ void stage_in(uint32_t* lds, const uint8_t* base, size_t block_first, size_t n) {
  const size_t byte0 = block_first * REC;                           // multiple of 256*REC: 4-byte aligned when base is
  const size_t total = n * (size_t)REC;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(base + byte0);
  for (int d = threadIdx.x; d < rec_lds<REC>::DWORDS; d += BLOCK) {
    const size_t off = byte0 + 4 * (size_t)d;
    uint32_t v = 0;
    if (off + 4 <= total) v = src[d];
    else for (int b = 0; b < 4; ++b) if (off + b < total) v |= (uint32_t)base[off + b] << (8 * b);   // ragged tail
    lds[d] = v;
  }
  __syncthreads();
}
template <int REC> __device__ void stage_out(const uint32_t* lds, uint8_t* base, size_t block_first, size_t n) {
  __syncthreads();
  const size_t byte0 = block_first * REC, total = n * (size_t)REC;
  uint32_t* dst = reinterpret_cast<uint32_t*>(base + byte0);
  for (int d = threadIdx.x; d < rec_lds<REC>::DWORDS; d += BLOCK) {
    const size_t off = byte0 + 4 * (size_t)d;
    if (off + 4 <= total) dst[d] = lds[d];
    else for (int b = 0; b < 4; ++b) if (off + b < total) base[off + b] = (uint8_t)(lds[d] >> (8 * b));
  }
}
__device__ __forceinline__ fe load_be32(const uint8_t* p) {            // 32 big-endian bytes -> words
  fe r;
#pragma unroll
  for (int w = 0; w < 8; ++w) { const uint8_t* q = p + 4 * (7 - w); r.w[w] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; }
  return r;
}
__device__ __forceinline__ void store_be32(uint8_t* p, const fe& v) {
#pragma unroll
  for (int w = 0; w < 8; ++w) { uint8_t* q = p + 4 * (7 - w); q[0] = (uint8_t)(v.w[w] >> 24); q[1] = (uint8_t)(v.w[w] >> 16); q[2] = (uint8_t)(v.w[w] >> 8); q[3] = (uint8_t)v.w[w]; }
}
template <int C> __device__ __forceinline__ bool below_p(const fe& v) {
  fe t = v; const fe p = FE_CONST(C, P);
  return sub8(t, p) != 0u;                                              // borrow <=> v < p
}

template <int C, int REC> __global__ void __launch_bounds__(BLOCK) k_sec1_encode(const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, uint8_t* __restrict__ out, size_t n) {
  __shared__ uint32_t lds[rec_lds<REC>::DWORDS];
  const size_t first = (size_t)blockIdx.x * BLOCK, i = first + threadIdx.x;
  uint8_t* rec = reinterpret_cast<uint8_t*>(lds) + threadIdx.x * REC;
  if (i < n) {
    const fe xv = fe_load(x, i), yv = fe_load(y, i);
    if constexpr (REC == 65) { rec[0] = 0x04; store_be32(rec + 1, xv); store_be32(rec + 33, yv); }
    else { rec[0] = (uint8_t)(0x02 | (yv.w[0] & 1u)); store_be32(rec + 1, xv); }
  }
  stage_out<REC>(lds, out, first, n);
}

template <int C, int REC> __global__ void __launch_bounds__(BLOCK) k_sec1_decode(const uint8_t* __restrict__ in, uint64_t* __restrict__ x, uint64_t* __restrict__ y, uint8_t* __restrict__ ok, size_t n) {
  __shared__ uint32_t lds[rec_lds<REC>::DWORDS];
  const size_t first = (size_t)blockIdx.x * BLOCK, i = first + threadIdx.x;
  stage_in<REC>(lds, in, first, n);
  if (i >= n) return;
  constexpr int CI = curve_domain<C>::fast;
  const uint8_t* rec = reinterpret_cast<const uint8_t*>(lds) + threadIdx.x * REC;
  const uint32_t prefix = rec[0];
  const fe xv = load_be32(rec + 1);
  const fe xf = classical_to_fast<C>(xv);
  fe rhs = fe_mul<CI>(fe_sqr<CI>(xf), xf);                              // x^3 + a x + b
  if constexpr (C == CURVE_P256) rhs = fe_sub<CI>(fe_add<CI>(rhs, FE_CONST(CI, BM)), fe_add<CI>(fe_dbl<CI>(xf), xf));
  else rhs = fe_add<CI>(rhs, FE_CONST(CI, BM));
  bool good = below_p<C>(xv);
  fe yv;
  if constexpr (REC == 65) {
    yv = load_be32(rec + 33);
    good = good && prefix == 0x04 && below_p<C>(yv) && fe_eq(fe_sqr<CI>(classical_to_fast<C>(yv)), rhs);
  } else {
    const fe s = fe_sqrt_candidate<CI>(rhs);              // p = 3 mod 4 (gfp.h:84)
    good = good && (prefix == 0x02 || prefix == 0x03) && fe_eq(fe_sqr<CI>(s), rhs);
    yv = fast_to_classical<C>(s);
    const fe neg = fe_neg<C>(yv);                                       // p - y (0 stays 0)
    const uint32_t flip = 0u - (uint32_t)((yv.w[0] & 1u) != (prefix & 1u));
    yv = fe_select(flip, neg, yv);
  }
  fe_store(x, i, xv); fe_store(y, i, yv);
  if (ok) ok[i] = (uint8_t)good;
}
// The same for a curve registered at run time (round 5): the generic field, y^2 = x^3 + a x + b with the curve's own a, p = 3 mod 4 like every registered curve.
template <int REC> __global__ void __launch_bounds__(BLOCK) k_gc_sec1_decode(gcurve G, const uint8_t* __restrict__ in, uint64_t* __restrict__ x, uint64_t* __restrict__ y, uint8_t* __restrict__ ok, size_t n) {
  __shared__ uint32_t lds[rec_lds<REC>::DWORDS];
  const size_t first = (size_t)blockIdx.x * BLOCK, i = first + threadIdx.x;
  stage_in<REC>(lds, in, first, n);
  if (i >= n) return;
  const uint8_t* rec = reinterpret_cast<const uint8_t*>(lds) + threadIdx.x * REC;
  const uint32_t prefix = rec[0];
  const fe xv = load_be32(rec + 1), P = g_words(G.F.p);
  const fe xm = g_from_classical(xv, G.F);
  const fe rhs = gc_add(gc_add(gc_mul(gc_sqr<false>(xm, G), xm, G), gc_mul(g_words(G.am), xm, G), G), g_words(G.bm), G);
  bool good = g_less(xv, P);
  fe yv;
  if constexpr (REC == 65) {
    yv = load_be32(rec + 33);
    good = good && prefix == 0x04 && g_less(yv, P) && fe_eq(gc_sqr<false>(g_from_classical(yv, G.F), G), rhs);
  } else {
    const fe s = gc_pow29(rhs, G.F.psqrt, G);
    good = good && (prefix == 0x02 || prefix == 0x03) && fe_eq(gc_sqr<false>(s, G), rhs);
    yv = g_to_classical(s, G.F);
    fe neg; (void)sub8_3(neg, P, yv);
    const uint32_t flip = (0u - (uint32_t)((yv.w[0] & 1u) != (prefix & 1u))) & ~g_zero_mask(yv);      // p - y (0 stays 0)
    yv = fe_select(flip, neg, yv);
  }
  fe_store(x, i, xv); fe_store(y, i, yv);
  if (ok) ok[i] = (uint8_t)good;
}
// Public-key validation (SEC 1 section 3.2.2.1 without the subgroup step: both curves have cofactor 1): ok[i] = x, y < p and
// y^2 = x^3 + a x + b.  (0, 0) -- this library's encoding of the point at infinity -- fails the equation (b != 0).
template <int C> __global__ void __launch_bounds__(BLOCK) k_on_curve(const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, uint8_t* __restrict__ ok, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  constexpr int CI = curve_domain<C>::fast;
  const fe xv = fe_load(x, i), yv = fe_load(y, i);
  const fe xf = classical_to_fast<C>(xv);
  fe rhs = fe_mul<CI>(fe_sqr<CI>(xf), xf);
  if constexpr (C == CURVE_P256) rhs = fe_sub<CI>(fe_add<CI>(rhs, FE_CONST(CI, BM)), fe_add<CI>(fe_dbl<CI>(xf), xf));
  else rhs = fe_add<CI>(rhs, FE_CONST(CI, BM));
  ok[i] = (uint8_t)(below_p<C>(xv) && below_p<C>(yv) && fe_eq(fe_sqr<CI>(classical_to_fast<C>(yv)), rhs));
}
// lanes whose input failed validation report the point at infinity: (0, 0), finite = 0
__global__ void __launch_bounds__(BLOCK) k_clear_invalid(const uint8_t* __restrict__ valid, uint64_t* __restrict__ rx, uint64_t* __restrict__ ry, uint8_t* __restrict__ finite, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n || valid[i]) return;
  fe z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z.w[j] = 0;
  fe_store(rx, i, z); if (ry) fe_store(ry, i, z); if (finite) finite[i] = 0;
}
}  // namespace

namespace launch {
#define GO(kern, ...) hipLaunchKernelGGL(kern, grid_for(n), dim3(BLOCK), 0, s, __VA_ARGS__)
void on_curve(hipStream_t s, int curve, const uint64_t* x, const uint64_t* y, uint8_t* ok, size_t n) {
  if (curve == CURVE_P256) GO((k_on_curve<CURVE_P256>), x, y, ok, n); else GO((k_on_curve<CURVE_SECP256K1>), x, y, ok, n);
}
void clear_invalid(hipStream_t s, const uint8_t* valid, uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n) { GO(k_clear_invalid, valid, rx, ry, finite, n); }
void bytes_be(hipStream_t s, const void* in, void* out, size_t n) { GO(k_bytes_be, static_cast<const uint4*>(in), static_cast<uint4*>(out), n); }
void mask_bit(hipStream_t s, const uint64_t* a, int bit, uint8_t* flag, size_t n) { GO(k_mask_bit, a, bit, flag, n); }
void wide4_to_lanes(hipStream_t s, const void* wides, size_t record_bytes, size_t offset_bytes, uint64_t* out, size_t n) {
  GO(k_wide4_to_lanes, static_cast<const uint8_t*>(wides), record_bytes, offset_bytes, reinterpret_cast<uint4*>(out), n); }
void lanes_to_wide4(hipStream_t s, const uint64_t* in, void* wides, size_t record_bytes, size_t offset_bytes, size_t n) {
  GO(k_lanes_to_wide4, reinterpret_cast<const uint4*>(in), static_cast<uint8_t*>(wides), record_bytes, offset_bytes, n); }
void sec1_encode(hipStream_t s, int curve, const uint64_t* x, const uint64_t* y, uint8_t* out, size_t n, bool compressed) {
  if (curve == CURVE_P256) { if (compressed) GO((k_sec1_encode<CURVE_P256, 33>), x, y, out, n); else GO((k_sec1_encode<CURVE_P256, 65>), x, y, out, n); }
  else { if (compressed) GO((k_sec1_encode<CURVE_SECP256K1, 33>), x, y, out, n); else GO((k_sec1_encode<CURVE_SECP256K1, 65>), x, y, out, n); }
}
void gc_sec1_decode(hipStream_t s, const gcurve& G, const uint8_t* in, uint64_t* x, uint64_t* y, uint8_t* ok, size_t n, bool compressed) {
  if (compressed) GO((k_gc_sec1_decode<33>), G, in, x, y, ok, n); else GO((k_gc_sec1_decode<65>), G, in, x, y, ok, n);
}
void sec1_decode(hipStream_t s, int curve, const uint8_t* in, uint64_t* x, uint64_t* y, uint8_t* ok, size_t n, bool compressed) {
  if (curve == CURVE_P256) { if (compressed) GO((k_sec1_decode<CURVE_P256, 33>), in, x, y, ok, n); else GO((k_sec1_decode<CURVE_P256, 65>), in, x, y, ok, n); }
  else { if (compressed) GO((k_sec1_decode<CURVE_SECP256K1, 33>), in, x, y, ok, n); else GO((k_sec1_decode<CURVE_SECP256K1, 65>), in, x, y, ok, n); }
}
}  // namespace launch
}  // namespace ecsimd_hip
