// k_gcomb.hip -- k G on a curve registered at RUN time from a table of multiples of its generator in LDS (round 5): k_affine.inc k_base_windowed
// (BASELINE configs[2]'s algorithm: 4-bit windows, 64 x 8 odd multiples (2d + 1) 16^w G, 32 KiB of LDS shared by the waves of a workgroup) with the
// curve's prime and order as kernel arguments.  A mixed addition (fe29.cuh madd29: Hankerson-Menezes-Vanstone Alg. 3.22, 8M + 3S) has no curve
// coefficient in it, so the loop is the built-in curves' loop with the dense reduction of gcurve.cuh -- tools/radix29_model.py prove_comb_invariant
// holds for every odd p < 2^256 (CURVE_ANY) as it stands.  Not the reference's algorithm: its results are compared as affine points (level A) with the
// ladder's, lane for lane (tests/test_gpu_curves.py); what it buys is u1 G of an ECDSA verification and k G of a signature on such a curve at a sixth
// of a ladder pass.
//
// Regular recoding with odd digits (Joye-Tunstall) exactly as the built-in kernel: k mod n (one conditional subtraction: the host registers the comb
// only for n >= 2^255), the odd one of it and n - it (the sign goes to the result), digit w = (nibble w | 1) - 16 where nibble w + 1 is even, top digit
// = nibble 63 | 1; the accumulator starts from the top entry and takes 63 mixed additions, none of which is exceptional except the last one at the one
// scalar k* = n - 2 (n mod 16) (when bit 4 of it is clear), where the kernel substitutes the table's k* G (tests/test_accumulator_models.py walks the
// accumulator for the registered curves' orders too).  CT: every lane reads all eight entries of a window (one LDS address per wave, a broadcast) and
// keeps its own under lane masks; no address, branch or EXEC mask depends on the scalar (tests/test_constant_time_isa.py).
#include "kernels.h"
#include "gcurve.cuh"
#include "../../include/ecsimd_hip.h"

namespace ecsimd_hip {
namespace {
constexpr int C = CURVE_GENERIC;
constexpr int GC_WINDOWS = launch::GCOMB_WINDOWS, GC_PER = launch::GCOMB_ENTRIES;
constexpr int GC_TABLE_WORDS = GC_WINDOWS * GC_PER * 16;       // 32 KiB
constexpr int GC_WBLOCK = 256;

// classical affine (x, y) from the ladder -> the loop's domain (x 2^261 mod p, canonical, 8 words): an entry is ready for to29() alone
__global__ void __launch_bounds__(256) k_gc_pack_table(gcurve G, const uint64_t* __restrict__ tx, const uint64_t* __restrict__ ty, uint32_t* __restrict__ table, int entries) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= entries) return;
  const r29_ctx<C>& cx = G.r29;
  const fe x = canon29<C>(enter29<C>(g_from_classical(fe_load(tx, e), G.F), cx), cx);
  const fe y = canon29<C>(enter29<C>(g_from_classical(fe_load(ty, e), G.F), cx), cx);
#pragma unroll
  for (int i = 0; i < 8; ++i) { table[e * 16 + i] = x.w[i]; table[e * 16 + 8 + i] = y.w[i]; }
}

ECS_DEV void entry_words(const uint4* e, fe& tx, fe& ty) {
  const uint4 q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3];
  tx.w[0] = q0.x; tx.w[1] = q0.y; tx.w[2] = q0.z; tx.w[3] = q0.w; tx.w[4] = q1.x; tx.w[5] = q1.y; tx.w[6] = q1.z; tx.w[7] = q1.w;
  ty.w[0] = q2.x; ty.w[1] = q2.y; ty.w[2] = q2.z; ty.w[3] = q2.w; ty.w[4] = q3.x; ty.w[5] = q3.y; ty.w[6] = q3.z; ty.w[7] = q3.w;
}
template <bool CT> ECS_DEV void window_entry(const uint4* win, uint32_t slot, fe& tx, fe& ty) {
  if constexpr (!CT) { entry_words(win + slot * 4, tx, ty); return; }
  entry_words(win, tx, ty);
#pragma unroll
  for (int e = 1; e < GC_PER; ++e) {
    fe ex, ey;
    entry_words(win + e * 4, ex, ey);
    const uint32_t m = 0u - (uint32_t)(slot == (uint32_t)e);
    tx = fe_select(m, ex, tx); ty = fe_select(m, ey, ty);
  }
}

template <bool CT> __global__ void __launch_bounds__(GC_WBLOCK)
k_gc_base_windowed(gcurve G, launch::words8 order8, const uint64_t* __restrict__ k, const uint32_t* __restrict__ table,
                   uint64_t* __restrict__ ox, uint64_t* __restrict__ oy, uint64_t* __restrict__ oz, size_t n) {
  __shared__ uint4 lds[GC_TABLE_WORDS / 4];
  {
    const uint4* src = reinterpret_cast<const uint4*>(table);
    for (int e = threadIdx.x; e < GC_TABLE_WORDS / 4; e += GC_WBLOCK) lds[e] = src[e];
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * GC_WBLOCK + threadIdx.x;
  if (i >= n) return;
  const r29_ctx<C>& cx = G.r29;
  fe kf = fe_load(k, i);
  fe order;
#pragma unroll
  for (int j = 0; j < 8; ++j) order.w[j] = order8.w[j];
  {
    fe d;
    const uint32_t borrow = sub8_3(d, kf, order);        // k < 2^256 <= 2n
    kf = fe_select(borrow, kf, d);
  }
  const uint32_t zmask = g_zero_mask(kf);
  const uint32_t flip = 0u - (uint32_t)((kf.w[0] & 1u) == 0u);
  {
    fe nk;
    (void)sub8_3(nk, order, kf);
    kf = fe_select(flip, nk, kf);
  }
  kf.w[0] = (zmask & 1u) | (kf.w[0] & ~zmask);           // k = 0 mod n: any odd value; replaced by infinity below
  // the one scalar whose last addition meets R = T: table[entries] = k* G, table[entries + 1] = {k* (8 words), 0...} (k* = 0: no such scalar)
  const uint4* tail = reinterpret_cast<const uint4*>(table) + ((size_t)GC_WINDOWS * GC_PER + 1) * 4;
  uint32_t special;
  {
    const uint4 a = tail[0], b = tail[1];
    const uint32_t d = (kf.w[0] ^ a.x) | (kf.w[1] ^ a.y) | (kf.w[2] ^ a.z) | (kf.w[3] ^ a.w) | (kf.w[4] ^ b.x) | (kf.w[5] ^ b.y) | (kf.w[6] ^ b.z) | (kf.w[7] ^ b.w);
    special = 0u - (uint32_t)(d == 0u);
  }
  const fe29 one = enter29<C>(g_words(G.F.r), cx);        // 2^261 mod p, tight: the field's 1 in the loop's domain
  jpoint29 A;
  uint32_t above = kf.w[7] >> 28;
  {
    fe tx, ty;
    window_entry<CT>(&lds[(63 * GC_PER) * 4], above >> 1, tx, ty);
    A.x = to29(tx); A.y = to29(ty); A.z = one;
  }
#pragma unroll 1
  for (int w = 62; w >= 0; --w) {
#pragma unroll
    for (int j = 7; j > 0; --j) kf.w[j] = __builtin_amdgcn_alignbit(kf.w[j], kf.w[j - 1], 28);       // the next nibble down moves to the top
    kf.w[0] <<= 4;
    const uint32_t nib = kf.w[7] >> 28;
    const uint32_t u = nib | 1u;
    const uint32_t neg = 0u - (uint32_t)((above & 1u) == 0u);
    const uint32_t mag = (neg & (16u - u)) | (~neg & u);
    above = nib;
    fe tx, ty;
    window_entry<CT>(&lds[(w * GC_PER) * 4], mag >> 1, tx, ty);
    A = madd29<C>(A, to29(tx), cneg29(neg, to29(ty)), cx);
  }
  if (CT || __builtin_amdgcn_ballot_w64(special != 0u) != 0ull) {                                     // constant time: taken by every wave
    fe tx, ty;
    entry_words(reinterpret_cast<const uint4*>(table) + (size_t)GC_WINDOWS * GC_PER * 4, tx, ty);
    A.x = select29(special, to29(tx), A.x); A.y = select29(special, to29(ty), A.y); A.z = select29(special, one, A.z);
  }
  fe X = leave29<C>(A.x, cx), Y = leave29<C>(A.y, cx), Z = leave29<C>(A.z, cx);                       // API Montgomery form (x 2^256 mod p, canonical)
  Y = fe_select(flip, g_opposite(Y, G.F), Y);
#pragma unroll
  for (int j = 0; j < 8; ++j) { X.w[j] &= ~zmask; Y.w[j] &= ~zmask; Z.w[j] &= ~zmask; }                   // k = 0 mod n: infinity (Z = 0)
  fe_store(ox, i, X); fe_store(oy, i, Y); fe_store(oz, i, Z);
}

// ---- WB-bit windows with odd digits, summed from the bottom (k_affine.inc k_base_windowed_s<WB, CT>): NW = ceil(256 / WB) windows x 2^(WB-1) odd multiples
// (2d + 1) 2^(WB w) G in LDS.  The odd one of k mod n and n - k; digit w = ((k >> WB w) mod 2^(WB+1) | 1) - 2^WB for every window but the top one, whose
// digit is what remains | 1.  No zero digit: the first entry starts the sum, nothing to skip.  The one scalar whose last addition can meet R = T is
// k* = n - 2 (n mod 2^(WB (NW - 1))); the table's tail holds k* G and {k*, 0} as the 4-bit comb's does (tests/test_accumulator_models.py walks the
// accumulator for the registered curves' orders).  Three shapes are launched:
//   <7, false, 1024>  ALG_WINDOWED_SIGNED: 37 x 64 entries, 148 KiB of LDS, ONE workgroup of 1 024 threads per CU (four waves per SIMD: 128 registers), 36 mixed
//                     additions instead of 63; public scalars (u1 G of a verification);
//   <20, false, 256>  ALG_WINDOWED_BIG: 13 x 2^19 entries = 436 MB in DEVICE memory (no LDS), 12 mixed additions behind 13 dependent random 64-byte reads; public scalars;
//   <5, true, 256>    ALG_WINDOWED | ALG_CONSTANT_TIME and k G of ecdsa_sign: 52 x 16 entries, 53 KB of LDS, three workgroups per CU, 51 additions; every entry
//                     of a window is read (one LDS address per wave, a broadcast) and the lane's own kept under masks -- no address, branch or EXEC mask
//                     depends on the scalar (tests/test_constant_time_isa.py).
template <int WB> struct gswin { static constexpr int WINDOWS = (256 + WB - 1) / WB, PER = 1 << (WB - 1); static constexpr size_t TABLE_WORDS = (size_t)WINDOWS * PER * 16; };
template <int PER, bool CT> ECS_DEV void gs_entry(const uint4* win, uint32_t slot, fe& tx, fe& ty) {
  if constexpr (!CT) { entry_words(win + slot * 4, tx, ty); return; }
  entry_words(win, tx, ty);
#pragma unroll
  for (int e = 1; e < PER; ++e) {
    fe ex, ey;
    entry_words(win + e * 4, ex, ey);
    const uint32_t m = 0u - (uint32_t)(slot == (uint32_t)e);
    tx = fe_select(m, ex, tx); ty = fe_select(m, ey, ty);
  }
}
template <int WB, bool CT, int BLK> __global__ void __launch_bounds__(BLK)
k_gc_base_windowed_s(gcurve G, launch::words8 order8, const uint64_t* __restrict__ k, const uint32_t* __restrict__ table,
                     uint64_t* __restrict__ ox, uint64_t* __restrict__ oy, uint64_t* __restrict__ oz, size_t n) {
  constexpr int NW = gswin<WB>::WINDOWS, PER = gswin<WB>::PER;
  constexpr bool IN_LDS = WB < 16;                        // 20-bit windows: 13 x 2^19 entries = 436 MB stay in device memory, 13 dependent random 64-byte reads per lane
  extern __shared__ uint4 lds_s[];
  if constexpr (IN_LDS) {
    const uint4* src = reinterpret_cast<const uint4*>(table);
    for (int e = threadIdx.x; e < gswin<WB>::TABLE_WORDS / 4; e += BLK) lds_s[e] = src[e];
    __syncthreads();
  }
  const size_t i = (size_t)blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const r29_ctx<C>& cx = G.r29;
  constexpr uint32_t FULL = 1u << WB;
  fe kf = fe_load(k, i);
  fe order;
#pragma unroll
  for (int j = 0; j < 8; ++j) order.w[j] = order8.w[j];
  {
    fe d;
    const uint32_t borrow = sub8_3(d, kf, order);        // k < 2^256 <= 2n
    kf = fe_select(borrow, kf, d);
  }
  const uint32_t zmask = g_zero_mask(kf);
  const uint32_t flip = 0u - (uint32_t)((kf.w[0] & 1u) == 0u);
  {
    fe nk;
    (void)sub8_3(nk, order, kf);
    kf = fe_select(flip, nk, kf);
  }
  kf.w[0] = (zmask & 1u) | (kf.w[0] & ~zmask);           // k = 0 mod n: any odd value; replaced by infinity below
  const uint4* gtab = reinterpret_cast<const uint4*>(table);
  const uint4* tail = gtab + ((size_t)NW * PER + 1) * 4;
  uint32_t special;
  {
    const uint4 a = tail[0], b = tail[1];
    const uint32_t d = (kf.w[0] ^ a.x) | (kf.w[1] ^ a.y) | (kf.w[2] ^ a.z) | (kf.w[3] ^ a.w) | (kf.w[4] ^ b.x) | (kf.w[5] ^ b.y) | (kf.w[6] ^ b.z) | (kf.w[7] ^ b.w);
    special = 0u - (uint32_t)(d == 0u);
  }
  uint32_t kk[9];
#pragma unroll
  for (int j = 0; j < 8; ++j) kk[j] = kf.w[j];
  kk[8] = 0;
  auto digit = [&](bool top, uint32_t& mag, uint32_t& neg) {          // the next window (top: wave-uniform), then shift; selects, no branch
    const uint32_t u = (kk[0] & (2u * FULL - 1u)) | 1u;
    const uint32_t sneg = 0u - (uint32_t)(u < FULL);
    const uint32_t smag = (sneg & (FULL - u)) | (~sneg & (u - FULL));
    neg = top ? 0u : sneg;
    mag = top ? (kk[0] | 1u) : smag;                                    // what remains: the positive top digit
#pragma unroll
    for (int j = 0; j < 8; ++j) kk[j] = __builtin_amdgcn_alignbit(kk[j + 1], kk[j], WB);
  };
  const fe29 one = enter29<C>(g_words(G.F.r), cx);
  jpoint29 A;
  {
    uint32_t mag, neg;
    digit(false, mag, neg);
    fe tx, ty;
    if constexpr (IN_LDS) gs_entry<PER, CT>(&lds_s[0], mag >> 1, tx, ty); else entry_words(gtab + (size_t)(mag >> 1) * 4, tx, ty);
    A.x = to29(tx); A.y = cneg29(neg, to29(ty)); A.z = one;
  }
#pragma unroll 1
  for (int w = 1; w < NW; ++w) {
    if constexpr (CT) asm volatile("" : "+s"(w));                       // the window counter stays a scalar register: the exit test is an s_cmp
    uint32_t mag, neg;
    digit(w + 1 >= NW, mag, neg);
    fe tx, ty;
    if constexpr (IN_LDS) gs_entry<PER, CT>(&lds_s[(size_t)w * PER * 4], mag >> 1, tx, ty); else entry_words(gtab + ((size_t)w * PER + (mag >> 1)) * 4, tx, ty);
    A = madd29<C>(A, to29(tx), cneg29(neg, to29(ty)), cx);
  }
  if (CT || __builtin_amdgcn_ballot_w64(special != 0u) != 0ull) {       // constant time: taken by every wave
    fe tx, ty;
    entry_words(gtab + (size_t)NW * PER * 4, tx, ty);
    A.x = select29(special, to29(tx), A.x); A.y = select29(special, to29(ty), A.y); A.z = select29(special, one, A.z);
  }
  fe X = leave29<C>(A.x, cx), Y = leave29<C>(A.y, cx), Z = leave29<C>(A.z, cx);
  Y = fe_select(flip, g_opposite(Y, G.F), Y);
#pragma unroll
  for (int j = 0; j < 8; ++j) { X.w[j] &= ~zmask; Y.w[j] &= ~zmask; Z.w[j] &= ~zmask; }
  fe_store(ox, i, X); fe_store(oy, i, Y); fe_store(oz, i, Z);
}
static_assert(gswin<20>::WINDOWS == launch::GCOMB20_WINDOWS && gswin<20>::PER == launch::GCOMB20_ENTRIES, "kernels.h");
static_assert(gswin<7>::WINDOWS == launch::GCOMB7_WINDOWS && gswin<7>::PER == launch::GCOMB7_ENTRIES && gswin<5>::WINDOWS == launch::GCOMB5_WINDOWS && gswin<5>::PER == launch::GCOMB5_ENTRIES, "kernels.h");
}  // namespace

namespace launch {
template <int WB, bool CT, int BLK> static void gcs_launch(hipStream_t s, const gcurve& G, const words8& order, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n) {
  constexpr size_t lds = WB < 16 ? gswin<WB>::TABLE_WORDS * 4 : 0;
  if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gc_base_windowed_s<WB, CT, BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_gc_base_windowed_s<WB, CT, BLK>), dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), lds, s, G, order, k, table, ox, oy, oz, n);
}
// bits = 7: the signed 7-bit comb (public scalars); bits = 5: the constant-time 5-bit comb (every entry of a window read); bits = 20: the comb in device memory (public)
void gc_base_windowed_s(hipStream_t s, const gcurve& G, const words8& order, int bits, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n) {
  if (bits == 7) gcs_launch<7, false, 1024>(s, G, order, k, table, ox, oy, oz, n);
  else if (bits == 20) gcs_launch<20, false, 256>(s, G, order, k, table, ox, oy, oz, n);
  else gcs_launch<5, true, 256>(s, G, order, k, table, ox, oy, oz, n);
}
void gc_pack_table(hipStream_t s, const gcurve& G, const uint64_t* tx, const uint64_t* ty, uint32_t* table, int entries) {
  hipLaunchKernelGGL(k_gc_pack_table, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, s, G, tx, ty, table, entries);
}
void gc_base_windowed(hipStream_t s, const gcurve& G, const words8& order, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, bool constant_time) {
  const dim3 grid((unsigned)((n + GC_WBLOCK - 1) / GC_WBLOCK));
  if (constant_time) hipLaunchKernelGGL(k_gc_base_windowed<true>, grid, dim3(GC_WBLOCK), 0, s, G, order, k, table, ox, oy, oz, n);
  else hipLaunchKernelGGL(k_gc_base_windowed<false>, grid, dim3(GC_WBLOCK), 0, s, G, order, k, table, ox, oy, oz, n);
}
}  // namespace launch
}  // namespace ecsimd_hip
