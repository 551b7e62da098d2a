// k_gcurve.hip -- point-formula kernels for a curve registered at RUN time (gcurve.cuh; reference layers L4 / L5: jacobian_curve_point.h,
// curve_group.h instantiated with any Curve type).  The curve travels as a kernel argument (gcurve, 512 bytes, wave-uniform -> SGPRs).
// REF = the reference's square() as written (ECSIMD_HIP_REF_SQUARE_COMPAT).  The ladder kernels are in k_gladder.hip (compiled in parallel).
#include "kernels.h"
#include "gcurve.cuh"

namespace ecsimd_hip {
namespace {
using launch::BLOCK;
#define GID size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= n) return
#define LD(p) fe_load(p, i)
#define ST(p, v) fe_store(p, i, v)

__global__ void __launch_bounds__(BLOCK) k_gc_from_affine(gcurve G, const uint64_t* x, const uint64_t* y, uint64_t* jx, uint64_t* jy, uint64_t* jz, size_t n) {
  GID; ST(jx, g_from_classical(LD(x), G.F)); ST(jy, g_from_classical(LD(y), G.F)); ST(jz, g_words(G.F.r));
}
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_gc_to_affine(gcurve G, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n) {
  GID; const gjpoint P{LD(jx), LD(jy), LD(jz)}; fe ax, ay;
  gc_to_affine<REF>(P, ax, ay, G); ST(x, ax); if (y != nullptr) ST(y, ay);
}
// y^2 = x^3 + a x + b (curve_group.h:43-58 hard-codes a = -3 as x^3 + b - 3x: the same value where a = -3, the right one elsewhere); p = 3 mod 4
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_gc_compute_y(gcurve G, const uint64_t* x, uint64_t* y, uint8_t* ok, size_t n) {
  GID;
  const fe xm = g_from_classical(LD(x), G.F);
  const fe rhs = gc_add(gc_add(gc_mul(gc_sqr<REF>(xm, G), xm, G), gc_mul(g_words(G.am), xm, G), G), g_words(G.bm), G);
  fe s;
  if constexpr (REF) s = g_pow<true>(rhs, G.F.psqrt, G.F);          // the reference's own power ladder, its squarings as written (mgry_ops.h:44-86)
  else s = gc_pow29(rhs, G.F.psqrt, G);
  if (ok) ok[i] = (uint8_t)fe_eq(gc_sqr<REF>(s, G), rhs);
  ST(y, g_to_classical(s, G.F));
}
// classical (x, y): x, y < p and on the curve ((0, 0), this library's point at infinity, fails unless b = 0 -- which no curve has)
__global__ void __launch_bounds__(BLOCK) k_gc_on_curve(gcurve G, const uint64_t* x, const uint64_t* y, uint8_t* ok, size_t n) {
  GID;
  const fe xc = LD(x), yc = LD(y), P = g_words(G.F.p);
  const fe xm = g_from_classical(xc, G.F), ym = g_from_classical(yc, G.F);
  const fe rhs = gc_add(gc_add(gc_mul(gc_sqr<false>(xm, G), xm, G), gc_mul(g_words(G.am), xm, G), G), g_words(G.bm), G);
  ok[i] = (uint8_t)(g_less(xc, P) && g_less(yc, P) && fe_eq(gc_sqr<false>(ym, G), rhs));
}
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_gc_dblu(gcurve G, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  GID; fe x = LD(px), y = LD(py), ox, oy, z;
  gc_dblu<REF>(x, y, ox, oy, z, G);
  const fe zo = z;
  ST(px, x); ST(py, y); ST(pz, zo); ST(rx, ox); ST(ry, oy); if (rz != pz) ST(rz, zo);
}
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_gc_zaddu(gcurve G, uint64_t* px, uint64_t* py, uint64_t* pz, const uint64_t* qx, const uint64_t* qy, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  GID; fe x = LD(px), y = LD(py), z = LD(pz), ox, oy;
  gc_zaddu<REF>(x, y, LD(qx), LD(qy), z, ox, oy, G);
  const fe zo = z;
  ST(px, x); ST(py, y); ST(pz, zo); ST(rx, ox); ST(ry, oy); if (rz != pz) ST(rz, zo);
}
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_gc_zdau(gcurve G, const uint64_t* px, const uint64_t* py, const uint64_t* pz, uint64_t* qx, uint64_t* qy, uint64_t* qz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  GID; fe x1 = LD(px), y1 = LD(py), z = LD(pz), x2 = LD(qx), y2 = LD(qy);
  gc_zdau<REF>(x1, y1, x2, y2, z, G);
  const fe zo = z;
  ST(qx, x2); ST(qy, y2); ST(qz, zo); ST(rx, x1); ST(ry, y1); if (rz != qz) ST(rz, zo);
}
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_gc_add_z2_1(gcurve G, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  GID; const gjpoint R = gc_add_z2_1<REF>(LD(ax), LD(ay), LD(az), LD(bx), LD(by), G);
  ST(rx, R.x); ST(ry, R.y); ST(rz, R.z);
}
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_gc_trplu(gcurve G, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  GID; fe x = LD(px), y = LD(py), dx2, dy2, z, ox, oy;
  gc_dblu<REF>(x, y, dx2, dy2, z, G);
  gc_zaddu<REF>(x, y, dx2, dy2, z, ox, oy, G);
  const fe zo = z;
  ST(px, x); ST(py, y); ST(pz, zo); ST(rx, ox); ST(ry, oy); if (rz != pz) ST(rz, zo);
}
// Jacobian (Montgomery form) -> affine classical with Montgomery's simultaneous inversion (k_affine.inc k_to_affine_batched for a registered curve):
// a lane owns m elements `lanes` apart, x[] holds the prefix products on the way up (x / y must not alias the inputs); Z = 0 -> (0, 0) like 0^(p-2).
__global__ void __launch_bounds__(256) k_gc_to_affine_batched(gcurve G, const uint64_t* __restrict__ jx, const uint64_t* __restrict__ jy, const uint64_t* __restrict__ jz,
                                                              uint64_t* __restrict__ x, uint64_t* __restrict__ y, size_t n, size_t lanes, int m) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= lanes) return;
  const fe one = g_words(G.F.r);
  fe acc = one;
  for (int j = 0; j < m; ++j) {
    const size_t e = (size_t)j * lanes + g;
    if (e >= n) break;
    fe z = fe_load(jz, e);
    z = fe_select(g_zero_mask(z), one, z);
    acc = gc_mul(acc, z, G);
    fe_store(x, e, acc);
  }
  fe inv = g_inverse_mgry(acc, G.F);
  int last = m - 1;
  while (last >= 0 && (size_t)last * lanes + g >= n) --last;
  for (int j = last; j >= 0; --j) {
    const size_t e = (size_t)j * lanes + g;
    fe z = fe_load(jz, e);
    const fe X = fe_load(jx, e);
    const uint32_t zero = g_zero_mask(z);
    z = fe_select(zero, one, z);
    const fe prev = (j > 0) ? fe_load(x, (size_t)(j - 1) * lanes + g) : one;
    fe iz = gc_mul(inv, prev, G);
    inv = gc_mul(inv, z, G);
#pragma unroll
    for (int k = 0; k < 8; ++k) iz.w[k] &= ~zero;
    const fe iz2 = gc_sqr<false>(iz, G);
    fe_store(x, e, g_to_classical(gc_mul(X, iz2, G), G.F));
    if (y != nullptr) fe_store(y, e, g_to_classical(gc_mul(fe_load(jy, e), gc_mul(iz2, iz, G), G), G.F));
  }
}
// Batched affine addition R = A + B (classical coordinates) for a registered curve: k_affine.inc k_affine_add_batched with the generic field -- one shared
// inversion per lane's m elements; (0, 0) is the point at infinity; finite[i] = 0 marks an infinite sum.  Public data (signature verification).
__global__ void __launch_bounds__(256) k_gc_affine_add_batched(gcurve G, const uint64_t* __restrict__ ax, const uint64_t* __restrict__ ay, const uint64_t* __restrict__ bx, const uint64_t* __restrict__ by,
                                                               uint64_t* __restrict__ rx, uint64_t* __restrict__ ry, uint8_t* __restrict__ finite, size_t n, size_t lanes, int m) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= lanes) return;
  const fe one = g_words(G.F.r);
  auto ld = [&](const uint64_t* p, size_t e) { return g_from_classical(fe_load(p, e), G.F); };
  fe acc = one;
  for (int j = 0; j < m; ++j) {
    const size_t e = (size_t)j * lanes + g;
    if (e >= n) break;
    const fe x1 = ld(ax, e), y1 = ld(ay, e), x2 = ld(bx, e), y2 = ld(by, e);
    const bool inf1 = g_is_zero(x1) && g_is_zero(y1), inf2 = g_is_zero(x2) && g_is_zero(y2);
    const bool same_x = fe_eq(x1, x2), same_y = fe_eq(y1, y2);
    fe den = gc_sub(x2, x1, G);
    if (same_x) den = gc_dbl(y1, G);
    if (inf1 || inf2 || (same_x && !same_y) || g_is_zero(den)) den = one;
    acc = gc_mul(acc, den, G);
    fe_store(rx, e, acc);
  }
  fe inv = g_inverse_mgry(acc, G.F);
  int last = m - 1;
  while (last >= 0 && (size_t)last * lanes + g >= n) --last;
  for (int j = last; j >= 0; --j) {
    const size_t e = (size_t)j * lanes + g;
    const fe x1 = ld(ax, e), y1 = ld(ay, e), x2 = ld(bx, e), y2 = ld(by, e);
    const bool inf1 = g_is_zero(x1) && g_is_zero(y1), inf2 = g_is_zero(x2) && g_is_zero(y2);
    const bool same_x = fe_eq(x1, x2), same_y = fe_eq(y1, y2);
    fe den = gc_sub(x2, x1, G), num = gc_sub(y2, y1, G);
    if (same_x) {                                              // tangent slope (3 x^2 + a) / (2 y)
      den = gc_dbl(y1, G);
      const fe xx = gc_sqr<false>(x1, G);
      num = gc_add(gc_add(gc_dbl(xx, G), xx, G), g_words(G.am), G);
    }
    const bool no_slope = inf1 || inf2 || (same_x && !same_y) || g_is_zero(den);
    if (no_slope) den = one;
    const fe prev = (j > 0) ? fe_load(rx, (size_t)(j - 1) * lanes + g) : one;
    const fe iden = gc_mul(inv, prev, G);
    inv = gc_mul(inv, den, G);
    const fe lam = gc_mul(num, iden, G);
    fe x3 = gc_sub(gc_sub(gc_sqr<false>(lam, G), x1, G), x2, G);
    fe y3 = gc_sub(gc_mul(lam, gc_sub(x1, x3, G), G), y1, G);
    bool fin = true;
    if (inf1) { x3 = x2; y3 = y2; fin = !inf2; }
    else if (inf2) { x3 = x1; y3 = y1; }
    else if (no_slope) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { x3.w[i] = 0; y3.w[i] = 0; }
      fin = false;
    }
    fe_store(rx, e, g_to_classical(x3, G.F));
    if (ry) fe_store(ry, e, g_to_classical(y3, G.F));
    if (finite) finite[e] = (uint8_t)fin;
  }
}
// ECDSA's acceptance test: ok = finite && x mod n == r, for x classical in [0, p) and p < 2n (checked by the caller: one conditional subtraction reduces x)
__global__ void __launch_bounds__(BLOCK) k_gc_x_mod_n_equals(gmod N, const uint64_t* __restrict__ x, const uint8_t* __restrict__ finite, const uint64_t* __restrict__ r, uint8_t* __restrict__ ok, size_t n) {
  GID; fe v = LD(x);
  g_cond_sub(v, 0, N);
  ok[i] = (uint8_t)(finite[i] != 0 && fe_eq(v, LD(r)));
}
// The reference's ladder (curve_group.h:189-218 as written) returns a meaningless point at three scalars: n - 1, 2^256 - n - 1 and 2^256 - n (the Joye ladder
// keeps R0 + R1 = 2^i P and meets n P = infinity inside a formula).  ECDSA on a registered curve multiplies through that ladder, so a
// scalar u < n that IS one of them is replaced by n - u here and the product's y negated afterwards (neg[i] = 1): u P = -((n - u) P).  The host has checked
// at registration that n - u is not degenerate in turn.  Selects only: the scalar may be a nonce.
__global__ void __launch_bounds__(BLOCK) k_gc_ladder_safe_scalars(gmod N, const uint64_t* __restrict__ u, uint64_t* __restrict__ adj, uint8_t* __restrict__ neg, size_t n) {
  GID; const fe k = LD(u), order = g_words(N.p);
  fe zero, one, nm1, c, cm1, alt;
#pragma unroll
  for (int j = 0; j < 8; ++j) { zero.w[j] = 0; one.w[j] = j == 0 ? 1u : 0u; }
  (void)sub8_3(nm1, order, one); (void)sub8_3(c, zero, order); (void)sub8_3(cm1, c, one);
  (void)sub8_3(alt, order, k);
  const uint32_t bad = 0u - (uint32_t)((int)fe_eq(k, nm1) | (int)fe_eq(k, c) | (int)fe_eq(k, cm1));
  ST(adj, fe_select(bad, alt, k));
  neg[i] = (uint8_t)(bad & 1u);
}
// y -> p - y where neg[i] (classical coordinates; (0, 0), the point at infinity, stays (0, 0))
__global__ void __launch_bounds__(BLOCK) k_gc_negate_where(gcurve G, const uint8_t* __restrict__ neg, uint64_t* __restrict__ y, size_t n) {
  GID; const fe v = LD(y);
  fe d; (void)sub8_3(d, g_words(G.F.p), v);
  const uint32_t m = (0u - (uint32_t)(neg[i] != 0)) & ~g_zero_mask(v);
  ST(y, fe_select(m, d, v));
}
#undef LD
#undef ST
}  // namespace

namespace launch {
#define GO(kern, ...) hipLaunchKernelGGL(kern, grid_for(n), dim3(BLOCK), 0, s, G, __VA_ARGS__)
#define GO_REF(kern, ...) do { if (ref) GO(kern<true>, __VA_ARGS__); else GO(kern<false>, __VA_ARGS__); } while (0)
void gc_from_affine(hipStream_t s, const gcurve& G, const uint64_t* x, const uint64_t* y, uint64_t* jx, uint64_t* jy, uint64_t* jz, size_t n) { GO(k_gc_from_affine, x, y, jx, jy, jz, n); }
void gc_to_affine(hipStream_t s, const gcurve& G, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n, bool ref) { GO_REF(k_gc_to_affine, jx, jy, jz, x, y, n); }
void gc_compute_y(hipStream_t s, const gcurve& G, const uint64_t* x, uint64_t* y, uint8_t* ok, size_t n, bool ref) { GO_REF(k_gc_compute_y, x, y, ok, n); }
void gc_on_curve(hipStream_t s, const gcurve& G, const uint64_t* x, const uint64_t* y, uint8_t* ok, size_t n) { GO(k_gc_on_curve, x, y, ok, n); }
void gc_dblu(hipStream_t s, const gcurve& G, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref) { GO_REF(k_gc_dblu, px, py, pz, rx, ry, rz, n); }
void gc_zaddu(hipStream_t s, const gcurve& G, uint64_t* px, uint64_t* py, uint64_t* pz, const uint64_t* qx, const uint64_t* qy, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref) { GO_REF(k_gc_zaddu, px, py, pz, qx, qy, rx, ry, rz, n); }
void gc_zdau(hipStream_t s, const gcurve& G, const uint64_t* px, const uint64_t* py, const uint64_t* pz, uint64_t* qx, uint64_t* qy, uint64_t* qz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref) { GO_REF(k_gc_zdau, px, py, pz, qx, qy, qz, rx, ry, rz, n); }
void gc_add_z2_1(hipStream_t s, const gcurve& G, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref) { GO_REF(k_gc_add_z2_1, ax, ay, az, bx, by, rx, ry, rz, n); }
void gc_trplu(hipStream_t s, const gcurve& G, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref) { GO_REF(k_gc_trplu, px, py, pz, rx, ry, rz, n); }
void gc_to_affine_batched(hipStream_t s, const gcurve& G, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n) {
  size_t m = n >> 17; if (m < 1) m = 1; if (m > 128) m = 128;
  const size_t lanes = (n + m - 1) / m;
  hipLaunchKernelGGL(k_gc_to_affine_batched, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, G, jx, jy, jz, x, y, n, lanes, (int)m);
}
void gc_affine_add_batched(hipStream_t s, const gcurve& G, const uint64_t* ax, const uint64_t* ay, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n) {
  size_t m = n >> 17; if (m < 1) m = 1; if (m > 128) m = 128;
  const size_t lanes = (n + m - 1) / m;
  hipLaunchKernelGGL(k_gc_affine_add_batched, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, G, ax, ay, bx, by, rx, ry, finite, n, lanes, (int)m);
}
void gc_x_mod_n_equals(hipStream_t s, const gmod& N, const uint64_t* x, const uint8_t* finite, const uint64_t* r, uint8_t* ok, size_t n) {
  hipLaunchKernelGGL(k_gc_x_mod_n_equals, grid_for(n), dim3(BLOCK), 0, s, N, x, finite, r, ok, n); }
void gc_ladder_safe_scalars(hipStream_t s, const gmod& N, const uint64_t* u, uint64_t* adj, uint8_t* neg, size_t n) {
  hipLaunchKernelGGL(k_gc_ladder_safe_scalars, grid_for(n), dim3(BLOCK), 0, s, N, u, adj, neg, n); }
void gc_negate_where(hipStream_t s, const gcurve& G, const uint8_t* neg, uint64_t* y, size_t n) { GO(k_gc_negate_where, neg, y, n); }
#undef GO
#undef GO_REF
}  // namespace launch
}  // namespace ecsimd_hip
