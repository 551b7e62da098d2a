// k_gvarwin.hip -- variable-base scalar multiplication with per-lane window tables on a curve registered at RUN time (round 5):
// ecsimd_hip_scalar_mult(registered curve, ALG_WINDOWED | OUT_AFFINE).  k_varwin.inc's odd-digit loop (P-256's) for ANY curve coefficient a and
// any prime, on fe29.cuh's nine signed 29-bit limbs with the dense prime in SGPRs.  Not the reference's algorithm (curve_group.h:189-218 has the
// co-Z ladder only): results are compared as affine points (level A) with the ladder's and the oracle's, lane for lane (tests/test_gpu_curves.py).
//
// Two kernels, nothing shared between lanes, no inversion before the final conversion:
//   k_gvw_table   the lane's eight odd multiples {1, 3, .., 15} P over ONE Z.  2P = gjdbl29(P); P over Z_2 (three products); (2j + 3) P = (2j + 1) P + 2P by
//                 co-Z additions (zaddu29, curve_group.h:91-116), each Z the one before times that step's x-difference -- so every multiple's Z divides
//                 the last one's, Zg, and (2j + 1) P = (X_j f_j^2, Y_j f_j^3) over Zg with f_j the product of the later differences: one walk back,
//                 five products per entry.  (x, y) -> (x Zg^2, y Zg^3) maps y^2 = x^3 + a x + b onto y^2 = x^3 + a Zg^4 x + b Zg^6, where those pairs
//                 are AFFINE points: the loop runs there, with a' = a Zg^4 as its coefficient (k_varwin.inc k_varwin_table_iso does this for a = 0).
//   k_gvw_mult    63 windows of three doublings and one fused double-add (2R + T = (R + T) + R, the second addition co-Z: fe29.cuh dbl_add29) over
//                 the odd digits of k or n - k (Joye-Tunstall: no zero digit, no point at infinity inside the loop).  The doublings are modified
//                 Jacobian (fe29.cuh gjdbl29): W = a' Z^4 is formed once per window (2S + 1M) and carried through the doublings (1M each), so a
//                 general a costs 25M + 19S per window where a Z^4 from Z every time would cost 27M + 23S.  A result (X', Y', Z') on the isomorphic
//                 curve is (X', Y', Z' Zg) on the curve itself: one product at the end.
// 63 x 44 + 66 (table) + 11 = 2 849 field multiplications against the ladder's 4 064 + 24.  The default loop indexes the table by the scalar's digits
// (PUBLIC scalars); k_gvw_mult<true> (ALG_CONSTANT_TIME) reads all eight entries in every window and keeps one under lane masks: secret scalars.
//
// Needs what the comb of the generator needs (capi.hip gc_comb_possible): the group order n, n >= 2^255 (k mod n by ONE conditional subtraction) -- and,
// like every table algorithm in this library, a group of prime order (no multiple (2j + 1) P, 2P, R, T of a point of order n coincides up to sign
// inside the loop; a curve with a cofactor keeps the ladder).  Interval proofs for every odd p < 2^256: tools/radix29_model.py prove_gwindow_invariant,
// prove_gtable; the device functions are tied to the model structurally (tools/fe29_structure.py).
#include "kernels.h"
#include "gcurve.cuh"
#include "../../include/ecsimd_hip.h"

namespace ecsimd_hip {
namespace {
using launch::BLOCK;
constexpr int C = CURVE_GENERIC;
constexpr int GVW_ENTRIES = 8;      // {1, 3, .., 15} P
constexpr int GVW_CHAIN = 7;        // co-Z additions

ECS_DEV void gvw_store_half(uint4* __restrict__ table, size_t slot, int half, const fe& v) {
  uint4* e = table + slot * 4 + 2 * half;
  e[0] = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
  e[1] = make_uint4(v.w[4], v.w[5], v.w[6], v.w[7]);
}
ECS_DEV void gvw_load_slot(const uint4* __restrict__ table, size_t slot, fe& x, fe& y) {
  const uint4* e = table + slot * 4;
  const uint4 q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3];
  x.w[0] = q0.x; x.w[1] = q0.y; x.w[2] = q0.z; x.w[3] = q0.w; x.w[4] = q1.x; x.w[5] = q1.y; x.w[6] = q1.z; x.w[7] = q1.w;
  y.w[0] = q2.x; y.w[1] = q2.y; y.w[2] = q2.z; y.w[3] = q2.w; y.w[4] = q3.x; y.w[5] = q3.y; y.w[6] = q3.z; y.w[7] = q3.w;
}

// table: 8 x 64 B per lane, entry j = (2j + 1) P over Zg as the canonical residues of x 2^261, y 2^261 (to29 gives tight limbs back at the read);
// zg: the lane's Zg, the same form.  The chain's waiting multiples and ratios live in the lane's own scratch (rolled loops, dynamic indices).
__global__ void __launch_bounds__(BLOCK, 2)
k_gvw_table(gcurve G, const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, int flags, uint4* __restrict__ table, uint64_t* __restrict__ zg, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const r29_ctx<C>& cx = G.r29;
  fe px = fe_load(x, i), py = fe_load(y, i);
  if (!(flags & ECSIMD_HIP_BASE_MGRY)) { px = g_from_classical(px, G.F); py = g_from_classical(py, G.F); }
  const fe29 x1 = enter29<C>(px, cx), y1 = enter29<C>(py, cx);
  const fe29 one = enter29<C>(g_words(G.F.r), cx);           // 2^261 mod p: the field's 1 in the loop's domain
  fe29 w = enter29<C>(g_words(G.am), cx);                    // a Z^4 at Z = 1
  jpoint29 Q; Q.x = x1; Q.y = y1; Q.z = one;
  Q = gjdbl29<C, false>(Q, w, cx);                           // 2P over Z_2
  fe29 z = Q.z;
  fe29 ax, ay;                                               // the running odd multiple, co-Z with 2P: P over Z_2 first
  { const fe29 zz = sqr29<C>(z, cx); ax = mul29<C>(x1, zz, cx); ay = mul29<C>(y1, mul29<C>(zz, z, cx), cx); }
  fe29 dx2 = Q.x, dy2 = Q.y;                                 // 2P, re-expressed over every new Z by zaddu29
  fe29 h[GVW_CHAIN], ex[GVW_CHAIN], ey[GVW_CHAIN];           // h[j] = Z^(j+1) / Z^(j); (ex, ey)[j] = (2j + 1) P as born, over Z^(j)
#pragma unroll 1
  for (int j = 0; j < GVW_CHAIN; ++j) {                      // (2j + 3) P = 2P + (2j + 1) P (never +-: a group of prime order)
    ex[j] = ax; ey[j] = ay;
    fe29 rx, ry, dx;
    zaddu29<C>(dx2, dy2, ax, ay, z, rx, ry, dx, cx);
    ax = rx; ay = ry;
    h[j] = dx;
  }
  const size_t base = i * GVW_ENTRIES;
  // 15 P over its own Z = Zg: a product with the field's 1 brings the lazy co-Z sum into canon29's domain
  gvw_store_half(table, base + GVW_CHAIN, 0, canon29<C>(mul29<C>(ax, one, cx), cx));
  gvw_store_half(table, base + GVW_CHAIN, 1, canon29<C>(mul29<C>(ay, one, cx), cx));
  fe_store(zg, i, canon29<C>(z, cx));
  fe29 f = h[GVW_CHAIN - 1];
#pragma unroll 1
  for (int j = GVW_CHAIN - 1; j >= 0; --j) {                 // f = Zg / Z^(j)
    const fe29 f2 = sqr29<C>(f, cx);
    gvw_store_half(table, base + j, 0, canon29<C>(mul29<C>(ex[j], f2, cx), cx));
    gvw_store_half(table, base + j, 1, canon29<C>(mul29<C>(ey[j], mul29<C>(f2, f, cx), cx), cx));
    if (j > 0) f = mul29<C>(f, h[j - 1], cx);
  }
}

#ifndef GVARWIN_WAVES_PER_SIMD
#define GVARWIN_WAVES_PER_SIMD 3
#endif
// CT (ALG_CONSTANT_TIME: secret scalars): every window reads ALL eight entries of the lane's table -- 512 contiguous bytes at an address made of the lane
// index alone -- and keeps the wanted one under lane masks, as k_varwin.inc k_varwin_mult_odd<true> does: four rounds of eight 16-byte loads (two whole
// entries, one 128-byte line), each in flight behind one of the window's four chunks of arithmetic; the first round of the NEXT window is requested before
// the double-add.  No address, no branch, no lane mask at a memory access depends on the scalar (tests/test_constant_time_isa.py).
template <bool CT> __global__ void __launch_bounds__(BLOCK, GVARWIN_WAVES_PER_SIMD)
k_gvw_mult(gcurve G, launch::words8 order8, const uint64_t* __restrict__ k, int k_stride, const uint4* __restrict__ table, const uint64_t* __restrict__ zg,
           uint64_t* __restrict__ ox, uint64_t* __restrict__ oy, uint64_t* __restrict__ oz, size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const r29_ctx<C>& cx = G.r29;
  fe kk = fe_load(k, k_stride ? i : 0);
  fe order;
#pragma unroll
  for (int j = 0; j < 8; ++j) order.w[j] = order8.w[j];
  {                                                     // k mod n (k < 2^256 <= 2n)
    fe d;
    const uint32_t borrow = sub8_3(d, kk, order);
    kk = fe_select(borrow, kk, d);
  }
  const uint32_t zmask = g_zero_mask(kk);
  const uint32_t flip = 0u - (uint32_t)((kk.w[0] & 1u) == 0u);     // even: use n - k (odd) and negate the result
  {
    fe nk;
    (void)sub8_3(nk, order, kk);
    kk = fe_select(flip, nk, kk);
  }
  kk.w[0] = (zmask & 1u) | (kk.w[0] & ~zmask);          // k = 0 mod n: any odd value; the result is replaced by infinity below
  const size_t base = i * GVW_ENTRIES;
  // a' = a Zg^4: the coefficient of the isomorphic curve the table's entries are affine points of
  fe29 ap;
  {
    const fe29 z2 = sqr29<C>(to29(fe_load(zg, i)), cx);
    ap = mul29<C>(enter29<C>(g_words(G.am), cx), sqr29<C>(z2, cx), cx);
  }
  jpoint29 R;
  uint32_t above = kk.w[7] >> 28;                       // the nibble above the current one (its low bit is the digit's sign)
  auto window = [&](fe29& wz) {                         // W = a' Z^4, then the first two doublings carry it along
    wz = mul29<C>(ap, sqr29<C>(sqr29<C>(R.z, cx), cx), cx);
  };
  if constexpr (CT) {
    const uint4* mine = table + base * 4;               // this lane's 8 entries: 32 x 16 bytes = four 128-byte lines
    uint4 q[8];
    auto request = [&](int pair) {                      // entries 2 pair, 2 pair + 1: ONE line, read once per window
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = mine[pair * 8 + j];
      __builtin_amdgcn_sched_barrier(0);               // the loads stay in front of the arithmetic they hide behind
    };
    auto keep = [&](int pair, uint32_t slot, fe& tx, fe& ty) {          // first pair: entry 0 is the default, the others replace it under their masks
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool m = (pair == 0 && h == 0) || slot == (uint32_t)(2 * pair + h);
        const uint4 a = q[4 * h], b = q[4 * h + 1], c = q[4 * h + 2], d = q[4 * h + 3];
        tx.w[0] = m ? a.x : tx.w[0]; tx.w[1] = m ? a.y : tx.w[1]; tx.w[2] = m ? a.z : tx.w[2]; tx.w[3] = m ? a.w : tx.w[3];
        tx.w[4] = m ? b.x : tx.w[4]; tx.w[5] = m ? b.y : tx.w[5]; tx.w[6] = m ? b.z : tx.w[6]; tx.w[7] = m ? b.w : tx.w[7];
        ty.w[0] = m ? c.x : ty.w[0]; ty.w[1] = m ? c.y : ty.w[1]; ty.w[2] = m ? c.z : ty.w[2]; ty.w[3] = m ? c.w : ty.w[3];
        ty.w[4] = m ? d.x : ty.w[4]; ty.w[5] = m ? d.y : ty.w[5]; ty.w[6] = m ? d.z : ty.w[6]; ty.w[7] = m ? d.w : ty.w[7];
      }
    };
    {                                                   // top digit = (k >> 252) | 1, positive
      const uint32_t slot = above >> 1;
      fe tx, ty;
      request(0); keep(0, slot, tx, ty); request(1); keep(1, slot, tx, ty);
      request(2); keep(2, slot, tx, ty); request(3); keep(3, slot, tx, ty);
      R.x = to29(tx); R.y = to29(ty); R.z = enter29<C>(g_words(G.F.r), cx);
    }
    request(0);
#pragma unroll 1
    for (int w = 62; w >= 0; --w) {
      asm volatile("" : "+s"(w));                       // the window counter stays a scalar register: the exit test is an s_cmp (tools/ct_check.py refuses vcc branches)
#pragma unroll
      for (int j = 7; j > 0; --j) kk.w[j] = __builtin_amdgcn_alignbit(kk.w[j], kk.w[j - 1], 28);   // kk <<= 4
      kk.w[0] <<= 4;
      const uint32_t nib = kk.w[7] >> 28;
      const uint32_t u = nib | 1u;
      const uint32_t neg = 0u - (uint32_t)((above & 1u) == 0u);
      const uint32_t slot = ((neg & (16u - u)) | (~neg & u)) >> 1;
      above = nib;
      fe tx, ty;
      fe29 wz;
      window(wz);
      keep(0, slot, tx, ty); request(1);
      R = gjdbl29<C, true>(R, wz, cx);
      keep(1, slot, tx, ty); request(2);
      R = gjdbl29<C, true>(R, wz, cx);
      keep(2, slot, tx, ty); request(3);
      R = gjdbl29<C, false>(R, wz, cx);
      keep(3, slot, tx, ty); request(0);                // the next window's first line (after the last window: read and dropped)
      R = dbl_add29<C>(R, to29(tx), cneg29(neg, to29(ty)), cx);
    }
  } else {
    {                                                   // top digit = (k >> 252) | 1, positive
      fe tx, ty;
      gvw_load_slot(table, base + (above >> 1), tx, ty);
      R.x = to29(tx); R.y = to29(ty); R.z = enter29<C>(g_words(G.F.r), cx);
    }
#pragma unroll 1
    for (int w = 62; w >= 0; --w) {
#pragma unroll
      for (int j = 7; j > 0; --j) kk.w[j] = __builtin_amdgcn_alignbit(kk.w[j], kk.w[j - 1], 28);   // kk <<= 4
      kk.w[0] <<= 4;
      const uint32_t nib = kk.w[7] >> 28;
      const uint32_t u = nib | 1u;                                        // 1, 3, ..., 15
      const uint32_t neg = 0u - (uint32_t)((above & 1u) == 0u);          // digit = u - 16 when the nibble above is even
      const uint32_t mag = neg ? 16u - u : u;
      above = nib;
      fe tx, ty;
      gvw_load_slot(table, base + (mag >> 1), tx, ty);                    // in flight during the doublings
      fe29 wz;
      window(wz);
      R = gjdbl29<C, true>(R, wz, cx);
      R = gjdbl29<C, true>(R, wz, cx);
      R = gjdbl29<C, false>(R, wz, cx);
      R = dbl_add29<C>(R, to29(tx), cneg29(neg, to29(ty)), cx);
    }
  }
  // back on the curve itself: Z = Z' Zg; API Montgomery form (x 2^256 mod p, canonical)
  fe X = leave29<C>(R.x, cx), Y = leave29<C>(R.y, cx), Z = leave29<C>(mul29<C>(R.z, to29(fe_load(zg, i)), cx), cx);
  Y = fe_select(flip, g_opposite(Y, G.F), Y);
#pragma unroll
  for (int j = 0; j < 8; ++j) { X.w[j] &= ~zmask; Y.w[j] &= ~zmask; Z.w[j] &= ~zmask; }                   // k = 0 mod n: infinity (Z = 0)
  fe_store(ox, i, X); fe_store(oy, i, Y); fe_store(oz, i, Z);
}
}  // namespace

namespace launch {
// scratch: gc_varwin_scratch_bytes(n), 32-byte aligned.  flags: ECSIMD_HIP_BASE_MGRY | ECSIMD_HIP_ALG_CONSTANT_TIME.  Affine classical (ox, oy) out; oy may be null.
void gc_varwin_scalar_mult(hipStream_t s, const gcurve& G, const words8& order, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, int flags,
                           uint64_t* scratch, uint64_t* ox, uint64_t* oy, size_t n) {
  uint4* table = reinterpret_cast<uint4*>(scratch);                      // 8n x 64 B
  uint64_t* zg = scratch + (size_t)GVW_ENTRIES * 8 * n;                    // n x 32 B
  uint64_t* jx = zg + 4 * n; uint64_t* jy = jx + 4 * n; uint64_t* jz = jy + 4 * n;
  hipLaunchKernelGGL(k_gvw_table, grid_for(n), dim3(BLOCK), 0, s, G, x, y, flags, table, zg, n);
  if (flags & ECSIMD_HIP_ALG_CONSTANT_TIME) hipLaunchKernelGGL(k_gvw_mult<true>, grid_for(n), dim3(BLOCK), 0, s, G, order, k, k_stride, (const uint4*)table, (const uint64_t*)zg, jx, jy, jz, n);
  else hipLaunchKernelGGL(k_gvw_mult<false>, grid_for(n), dim3(BLOCK), 0, s, G, order, k, k_stride, (const uint4*)table, (const uint64_t*)zg, jx, jy, jz, n);
  gc_to_affine_batched(s, G, jx, jy, jz, ox, oy, n);
}
}  // namespace launch
}  // namespace ecsimd_hip
