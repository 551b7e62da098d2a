// gcurve.cuh -- the reference's point layer for a curve given at RUN time (round 5), one point per lane.
//
// The reference's curve_group<Curve> (curve_group.h:20-255) is a template over ANY type with bn_type, P, A, B, Gx, Gy (curve.h:12-15):
// Am and Bm are derived from the type (curve_group.h:31-32), DBLU takes `a` from it (:64-87), ZADDU / ZDAU / ADD_Z2_1 / TRPLU and the
// ladder (:91-218) are curve-independent, and GFp<WBN, P> underneath needs p = 3 mod 4 (gfp.h:84).  point.cuh serves P-256 and secp256k1
// with their special-form field arithmetic; this header serves every other curve -- brainpoolP256r1, SM2, FRP256v1, GOST, ... -- from a
// `gcurve` record the host derives once per curve (capi.hip curve registry) and passes BY VALUE as a kernel argument: wave-uniform, so it
// lives in SGPRs.
//   * the formulas run on gfield.cuh's generic canonical arithmetic (8 x 32-bit words, word-serial Montgomery reduction by the dense p);
//   * the ladder's 254 iterations run on fe29.cuh's nine signed 29-bit limbs like the built-in curves' -- zdau29<CURVE_GENERIC>: the SAME
//     function, with the reduction's multiplier taken from the context (q_k = column * (-p^-1 mod 2^29), then + q_k * p over all nine limbs of
//     p: 81 multiply-adds per reduction where P-256's sparse form has 36; tools/radix29_model.py proves the loop invariant for EVERY odd
//     p < 2^256 at once, CURVE_ANY);
//   * REF: the reference's square() as written (mul.h:160-212, dropped carry included), for ECSIMD_HIP_REF_SQUARE_COMPAT -- on the 8-word loop.
// Every function returns canonical residues, so X, Y, Z are bit-identical to the reference instantiated with the same Curve (level J).
#pragma once
#include "gfield.cuh"
#include "fe29.cuh"

namespace ecsimd_hip {

struct gcurve {
  gmod F;                          // the field: p and what mgry_constants<WBN, P> derives from it
  uint32_t am[8], bm[8];           // a R mod p, b R mod p                                   curve_group.h:31-32
  uint32_t gx[8], gy[8];           // the generator, classical                               curve.h:12-15 (Gx, Gy)
  r29_ctx<CURVE_GENERIC> r29;      // the ladder loop's constants (fe29.cuh)
};

struct gjpoint { fe x, y, z; };

template <bool REF> ECS_DEV fe gc_sqr(const fe& a, const gcurve& G) { return g_sqr<REF>(a, G.F); }
ECS_DEV fe gc_mul(const fe& a, const fe& b, const gcurve& G) { return g_mul(a, b, G.F); }
ECS_DEV fe gc_add(const fe& a, const fe& b, const gcurve& G) { return g_add(a, b, G.F); }
ECS_DEV fe gc_sub(const fe& a, const fe& b, const gcurve& G) { return g_sub(a, b, G.F); }
ECS_DEV fe gc_dbl(const fe& a, const gcurve& G) { return g_dbl(a, G.F); }
template <int N> ECS_DEV fe gc_shl(fe a, const gcurve& G) {                       // mgry_ops.h:14-22: N successive doublings
#pragma unroll
  for (int i = 0; i < N; ++i) a = g_dbl(a, G.F);
  return a;
}

// curve_group.h:64-87 DBLU: P = (x, y, Z = R mod p).  Returns 2P in (rx, ry), rewrites (x, y) so that P and 2P share z.
template <bool REF> ECS_DEV void gc_dblu(fe& x, fe& y, fe& rx, fe& ry, fe& z, const gcurve& G) {
  const fe B = gc_sqr<REF>(x, G);
  const fe E = gc_sqr<REF>(y, G);
  const fe L = gc_sqr<REF>(E, G);
  fe t = gc_sqr<REF>(gc_add(x, E, G), G);
  t = gc_sub(gc_sub(t, B, G), L, G);
  const fe S = gc_dbl(t, G);
  const fe M = gc_add(gc_add(gc_dbl(B, G), B, G), g_words(G.am), G);             // 3B + a (curve_group.h:73: Am from the curve type)
  rx = gc_sub(gc_sqr<REF>(M, G), gc_dbl(S, G), G);
  const fe Lm8 = gc_shl<3>(L, G);
  ry = gc_sub(gc_mul(M, gc_sub(S, rx, G), G), Lm8, G);
  z = gc_dbl(y, G);
  x = S;
  y = Lm8;
}
// curve_group.h:91-116 ZADDU
template <bool REF> ECS_DEV void gc_zaddu(fe& x1, fe& y1, const fe& x2, const fe& y2, fe& z, fe& rx, fe& ry, const gcurve& G) {
  const fe dx = gc_sub(x1, x2, G);
  const fe Cc = gc_sqr<REF>(dx, G);
  const fe W1 = gc_mul(x1, Cc, G);
  const fe W2 = gc_mul(x2, Cc, G);
  const fe dy = gc_sub(y1, y2, G);
  const fe D = gc_sqr<REF>(dy, G);
  const fe A1 = gc_mul(y1, gc_sub(W1, W2, G), G);
  rx = gc_sub(gc_sub(D, W1, G), W2, G);
  ry = gc_sub(gc_mul(dy, gc_sub(W1, rx, G), G), A1, G);
  z = gc_mul(z, dx, G);
  x1 = W1;
  y1 = A1;
}
// curve_group.h:120-153 ZDAU: (x1, y1) <- 2 (x1, y1) + (x2, y2); (x2, y2) re-expressed with the new z.  The field values of point.cuh zdau<C>
// (shared sub-expressions taken once); `oswap` exchanges the two output points (the ladder folds its per-bit swaps into it).
template <bool REF> ECS_DEV void gc_zdau(fe& x1, fe& y1, fe& x2, fe& y2, fe& z, const gcurve& G, uint32_t oswap = 0u) {
  const fe dx = gc_sub(x1, x2, G);
  const fe Cp = gc_sqr<REF>(dx, G);
  const fe W1p = gc_mul(x1, Cp, G);
  const fe W2p = gc_mul(x2, Cp, G);
  const fe dy = gc_sub(y1, y2, G);
  const fe Dp = gc_sqr<REF>(dy, G);
  const fe A1p = gc_mul(y1, gc_sub(W1p, W2p, G), G);
  const fe X3p = gc_sub(gc_sub(Dp, W1p, G), W2p, G);
  const fe u = gc_sub(X3p, W1p, G);
  const fe Cc = gc_sqr<REF>(u, G);
  const fe A1p2 = gc_dbl(A1p, G);
  fe yp = gc_sub(gc_sub(gc_sqr<REF>(gc_sub(dy, u, G), G), Dp, G), Cc, G);       // Y3' + 2 A1'
  const fe Y3p = gc_sub(yp, A1p2, G);
  const fe C4 = gc_shl<2>(Cc, G);
  const fe W1 = gc_mul(X3p, C4, G);
  const fe W2 = gc_mul(W1p, C4, G);
  fe ym = gc_sub(Y3p, A1p2, G);
  const fe A1 = gc_mul(Y3p, gc_sub(W1, W2, G), G);
  const fe W12 = gc_add(W1, W2, G);
  fe zz = gc_sqr<REF>(gc_add(dx, u, G), G);
  zz = gc_sub(gc_sub(zz, Cp, G), Cc, G);
  z = gc_mul(z, zz, G);
  fe_cswap(oswap, ym, yp);
  const fe D = gc_sqr<REF>(ym, G);
  x1 = gc_sub(D, W12, G);
  y1 = gc_sub(gc_mul(ym, gc_sub(W1, x1, G), G), A1, G);
  const fe Dc = gc_sqr<REF>(yp, G);
  x2 = gc_sub(Dc, W12, G);
  y2 = gc_sub(gc_mul(yp, gc_sub(W1, x2, G), G), A1, G);
}
// curve_group.h:155-179 ADD_Z2_1: (X1, Y1, Z1) + affine (x2, y2) [Z2 = R mod p]
template <bool REF> ECS_DEV gjpoint gc_add_z2_1(const fe& X1, const fe& Y1, const fe& Z1, const fe& x2, const fe& y2, const gcurve& G) {
  const fe Z1Z1 = gc_sqr<REF>(Z1, G);
  const fe U2 = gc_mul(x2, Z1Z1, G);
  const fe S2 = gc_mul(gc_mul(y2, Z1, G), Z1Z1, G);
  const fe H = gc_sub(U2, X1, G);
  const fe HH = gc_sqr<REF>(H, G);
  const fe I = gc_shl<2>(HH, G);
  const fe J = gc_mul(H, I, G);
  const fe r = gc_dbl(gc_sub(S2, Y1, G), G);
  const fe V = gc_mul(X1, I, G);
  gjpoint R;
  R.x = gc_sub(gc_sub(gc_sqr<REF>(r, G), J, G), gc_dbl(V, G), G);
  R.y = gc_sub(gc_mul(r, gc_sub(V, R.x, G), G), gc_dbl(gc_mul(Y1, J, G), G), G);
  R.z = gc_sub(gc_sub(gc_sqr<REF>(gc_add(Z1, H, G), G), Z1Z1, G), HH, G);
  return R;
}
// gfp.h:42-44 inverse(): a^(p-2); for the prime p of a curve the division steps give the same unique inverse.  REF: the reference's own power ladder
// (its squarings as written).
template <bool REF> ECS_DEV fe gc_inverse(const fe& a, const gcurve& G) {
  if constexpr (REF) return g_pow<true>(a, G.F.pm2, G.F); else return g_inverse_mgry(a, G.F);
}
// jacobian_curve_point.h:33-42 to_affine: classical (x, y)
template <bool REF> ECS_DEV void gc_to_affine(const gjpoint& P, fe& ax, fe& ay, const gcurve& G) {
  const fe invZ = gc_inverse<REF>(P.z, G);
  const fe invZ2 = gc_sqr<REF>(invZ, G);
  const fe invZ3 = gc_mul(invZ2, invZ, G);
  ax = g_to_classical(gc_mul(P.x, invZ2, G), G.F);
  ay = g_to_classical(gc_mul(P.y, invZ3, G), G.F);
}

// a^e for a public, wave-uniform exponent (gfp.h:46-54 sqrt() = a^((p + 1) / 4) is the caller) on 3b's 29-bit limbs: the power of a canonical residue does
// not depend on how the exponent is walked (point.cuh fe_sqrt_candidate29 does the same with fixed chains for the built-in primes), so this walks it from
// the top in sliding windows of three bits over {a, a^3, a^5, a^7} -- 254 squarings + ~64 products of 126 / 162 multiply-adds where g_pow's canonical
// words spend ~450 / ~520 instructions on each of 254 + ~127.  Control flow is made of the exponent's bits alone (SGPRs).  Every factor is the output of
// a product (tight limbs), as in the ladder's loop.
ECS_DEV fe gc_pow29(const fe& a, const uint32_t (&e)[8], const gcurve& G) {
  constexpr int C = CURVE_GENERIC;
  const r29_ctx<C>& cx = G.r29;
  int i = -1;
  for (int t = 255; t >= 0; --t) if ((e[t >> 5] >> (t & 31)) & 1u) { i = t; break; }
  if (i < 0) return g_words(G.F.r);                                  // a^0
  const fe29 x1 = enter29<C>(a, cx);
  const fe29 x2 = sqr29<C>(x1, cx);
  const fe29 x3 = mul29<C>(x2, x1, cx), x5 = mul29<C>(x3, x2, cx), x7 = mul29<C>(x5, x2, cx);
  auto bit = [&](int t) -> uint32_t { return t < 0 ? 0u : (e[t >> 5] >> (t & 31)) & 1u; };
  fe29 r = x1;
  bool started = false;
#pragma unroll 1
  while (i >= 0) {
    if (!bit(i)) { r = sqr29<C>(r, cx); --i; continue; }
    int len = 3;                                                     // the longest window of at most three bits that ends in a one
    while (len > 1 && (i - len + 1 < 0 || !bit(i - len + 1))) --len;
    uint32_t val = 0;
    for (int t = 0; t < len; ++t) val = (val << 1) | bit(i - t);
    const fe29 f = val == 1u ? x1 : val == 3u ? x3 : val == 5u ? x5 : x7;
    if (started) {
      for (int t = 0; t < len; ++t) r = sqr29<C>(r, cx);
      r = mul29<C>(r, f, cx);
    } else { r = f; started = true; }
    i -= len;
  }
  return leave29<C>(r, cx);
}

// ---------------------------------------------------------------- the ladder (curve_group.h:189-218)
// TRPLU, the opening swaps and the even-k correction on canonical words as point.cuh ladder_core29; the 254 iterations on nine signed 29-bit limbs.
ECS_DEV uint32_t gc_ladder_core29(const uint32_t* __restrict__ kwords, const fe& xm, const fe& ym, fe& px, fe& py, fe& z, const gcurve& G) {
  fe bx, by;
  px = xm; py = ym;
  {
    fe dx2, dy2;
    gc_dblu<false>(px, py, dx2, dy2, z, G);
    gc_zaddu<false>(px, py, dx2, dy2, z, bx, by, G);
  }
  uint32_t kw = kwords[0];
  const uint32_t k0 = kw;
  uint32_t cur = 0u - ((kw >> 2) & 1u);
  {
    const uint32_t m = (0u - ((kw >> 1) & 1u)) ^ cur;
    fe_cswap(m, px, bx);
    fe_cswap(m, py, by);
  }
  constexpr int C = CURVE_GENERIC;
  const r29_ctx<C>& cx = G.r29;
  coz29 s;
  s.x1 = enter29<C>(bx, cx); s.x2 = enter29<C>(px, cx); s.y1 = enter29<C>(by, cx); s.z = enter29<C>(z, cx);
  s.dx = sub29(s.x1, s.x2);
  s.dy = sub29(s.y1, enter29<C>(py, cx));
#pragma unroll 1
  for (int b = 2; b < 256; ++b) {
    const int nb = b + 1;
    if ((nb & 31) == 0) kw = (nb < 256) ? kwords[nb >> 5] : 0u;
    const uint32_t next = 0u - ((kw >> (nb & 31)) & 1u);
    zdau29<C>(s, cur ^ next, cx);
    cur = next;
  }
  px = leave29<C>(s.x2, cx); py = leave29<C>(sub29(s.y1, s.dy), cx); z = leave29<C>(s.z, cx);
  return k0;
}
// ... and on gfield.cuh's canonical words (LADDER_RADIX32: the A/B of the dense reduction; REF: the reference's squaring)
template <bool REF> ECS_DEV uint32_t gc_ladder_core32(const uint32_t* __restrict__ kwords, const fe& xm, const fe& ym, fe& px, fe& py, fe& z, const gcurve& G) {
  fe bx, by;
  px = xm; py = ym;
  {
    fe dx2, dy2;
    gc_dblu<REF>(px, py, dx2, dy2, z, G);
    gc_zaddu<REF>(px, py, dx2, dy2, z, bx, by, G);
  }
  uint32_t kw = kwords[0];
  const uint32_t k0 = kw;
  uint32_t cur = 0u - ((kw >> 2) & 1u);
  {
    const uint32_t m = (0u - ((kw >> 1) & 1u)) ^ cur;
    fe_cswap(m, px, bx);
    fe_cswap(m, py, by);
  }
#pragma unroll 1
  for (int b = 2; b < 256; ++b) {
    const int nb = b + 1;
    if ((nb & 31) == 0) kw = (nb < 256) ? kwords[nb >> 5] : 0u;
    const uint32_t next = 0u - ((kw >> (nb & 31)) & 1u);
    gc_zdau<REF>(bx, by, px, py, z, G, cur ^ next);
    cur = next;
  }
  return k0;
}
template <int RADIX, bool REF> ECS_DEV gjpoint gc_scalar_mult_ladder(const uint32_t* __restrict__ kwords, const fe& xm, const fe& ym, const gcurve& G) {
  fe px, py, z;
  uint32_t k0;
  if constexpr (RADIX == 29) k0 = gc_ladder_core29(kwords, xm, ym, px, py, z, G);
  else k0 = gc_ladder_core32<REF>(kwords, xm, ym, px, py, z, G);
  // even k: subtract the original point once (curve_group.h:214-217)
  const gjpoint Psub = gc_add_z2_1<REF>(px, py, z, xm, g_opposite(ym, G.F), G);
  const uint32_t meven = 0u - (uint32_t)((k0 & 1u) == 0u);
  gjpoint R;
  R.x = fe_select(meven, Psub.x, px);
  R.y = fe_select(meven, Psub.y, py);
  R.z = fe_select(meven, Psub.z, z);
  return R;
}

}  // namespace ecsimd_hip
