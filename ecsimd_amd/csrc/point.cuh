// point.cuh -- co-Z Jacobian point formulas and the scalar-multiplication ladder, one point per lane.
//
// Replaces include/ecsimd/curve_group.h of the reference (formulas from ePrint 2010/309).  All
// coordinates are Montgomery-form canonical residues, so every function returns bit-identical
// X, Y, Z to the reference's (parity level J).  Where a sub-expression of the
// reference occurs twice it is computed once (u = X3'-W1', W1+W2): the VALUES are unchanged.
#pragma once
#include "field.cuh"
#include "fe29.cuh"

namespace ecsimd_hip {

// Two points sharing one Z (the reference asserts P.z == Q.z, curve_group.h:92,121).
struct coz_pair {
  fe x1, y1;   // first point  (the one ZDAU doubles / "base")
  fe x2, y2;   // second point (the one that is updated)
  fe z;
};

struct jpoint { fe x, y, z; };

// curve_group.h:64-87 DBLU: P = (x, y, Z = R mod p) affine.  Returns 2P in (rx, ry), rewrites
// (x, y) so that P and 2P share z.
template <int C> ECS_DEV void dblu(fe& x, fe& y, fe& rx, fe& ry, fe& z) {
  const fe B = fe_sqr<C>(x);
  const fe E = fe_sqr<C>(y);
  const fe L = fe_sqr<C>(E);
  fe t = fe_sqr<C>(fe_add<C>(x, E));
  t = fe_sub<C>(fe_sub<C>(t, B), L);
  const fe S = fe_dbl<C>(t);
  fe M = fe_add<C>(fe_dbl<C>(B), B);
  if constexpr (!curve_prime<C>::is_p256) { /* secp256k1, a = 0: M = 3B + 0 */ } else { M = fe_add<C>(M, FE_CONST(C, AM)); }
  rx = fe_sub<C>(fe_sqr<C>(M), fe_dbl<C>(S));
  const fe Lm8 = fe_shl<C, 3>(L);
  ry = fe_sub<C>(fe_mul<C>(M, fe_sub<C>(S, rx)), Lm8);
  z = fe_dbl<C>(y);
  x = S;
  y = Lm8;
}

// curve_group.h:91-116 ZADDU: returns (x1,y1)+(x2,y2) in (rx, ry), rewrites (x1, y1) and z so that
// all share the new z.
template <int C> ECS_DEV void zaddu(fe& x1, fe& y1, const fe& x2, const fe& y2, fe& z, fe& rx, fe& ry) {
  const fe dx = fe_sub<C>(x1, x2);
  const fe Cc = fe_sqr<C>(dx);
  const fe W1 = fe_mul<C>(x1, Cc);
  const fe W2 = fe_mul<C>(x2, Cc);
  const fe dy = fe_sub<C>(y1, y2);
  const fe D = fe_sqr<C>(dy);
  const fe A1 = fe_mul<C>(y1, fe_sub<C>(W1, W2));
  rx = fe_sub<C>(fe_sub<C>(D, W1), W2);
  ry = fe_sub<C>(fe_mul<C>(dy, fe_sub<C>(W1, rx)), A1);
  z = fe_mul<C>(z, dx);
  x1 = W1;
  y1 = A1;
}

// curve_group.h:120-153 ZDAU: (x1,y1) <- 2*(x1,y1) + (x2,y2); (x2,y2) re-expressed with the new z.
// 9M + 7S.  Same field values as the reference's DAG, with three things computed once instead of twice:
//   * u = X3' - W1' and W1 + W2 (shared sub-expressions);
//   * 4*X3'*C and 4*W1'*C quadruple C once;
//   * Y3' + 2A1' (the reference's curve_group.h:146) IS (dy + W1' - X3')^2 - D' - C: the term it adds back is the
//     one Y3' just subtracted, and both are canonical residues of the same element -- so it is taken, not recomputed.
// `oswap` (a per-lane all-ones / all-zeros word) exchanges the two OUTPUT points: (x1,y1) and (x2,y2) are
// (ym^2 - W12, ym*(W1 - x) - A1) and the same with yp, so swapping the outputs is swapping (ym, yp) -- one
// field-element swap instead of two.  The ladder folds its per-bit swaps into it; the point kernel passes 0.
// NOZ: the Z update (1M + 1S + 3 linear operations of the 9M + 7S) is left out -- the x-only ladder below does not need it.
// Where ZDAU updates Z (round 3).  "Early" = as soon as C = u^2 exists, so that dx and C' die before W1, W2 and the 16-word A1:
// same instruction count (3 400 / 3 471 VALU per iteration), shorter live ranges -- the P-256 ladder allocates 138 VGPRs instead
// of 147, the secp256k1 one 162 with NO spill instead of 168 with 14 spilled (60 B of scratch per lane, all but the scalar
// pointer's reload outside the bit loop).  Measured on one box, 2^24 lanes (profiles/r03/ab_zdau_z_placement_*.txt): P-256 48.29
// (early) against 48.45 M/s (late); secp256k1 48.33 against 48.41 -- registers are not what limits the ladder (the VALU is
// saturated at 3 waves per SIMD either way), so P-256 keeps the late placement and secp256k1 takes the spill-free one.
// -DECS_ZDAU_Z_EARLY=0 / 1 forces one placement for both curves (the A/B builds).
#ifndef ECS_ZDAU_Z_EARLY
#define ECS_ZDAU_Z_EARLY (-1)
#endif
template <int C> struct zdau_z_early { static constexpr bool value = (ECS_ZDAU_Z_EARLY < 0) ? !curve_prime<C>::is_p256 : (ECS_ZDAU_Z_EARLY != 0); };
// ZE: -1 = the curve's placement above, 0 / 1 = late / early for this call site; 2 = early with the factors of z * zz exchanged, which
// k_zdau_repeat<32> takes: with z as the FIRST factor, updated in place from registers a global load had pinned, the compiler handed
// mac_col's early-clobber carry counter the register of the still-live z.w[7] (a compiler defect, not a source one: the constraint is
// "=&v"; tools/asm_clobber_check.py finds it in the ISA and tests/test_isa_guards.py keeps every shipped unit checked).
template <int C, bool NOZ = false, int ZE = -1> ECS_DEV void zdau(fe& x1, fe& y1, fe& x2, fe& y2, fe& z, uint32_t oswap = 0u) {
  constexpr bool z_early = (ZE < 0) ? zdau_z_early<C>::value : (ZE != 0);
  const fe dx = fe_sub<C>(x1, x2);
  const fe Cp = fe_sqr<C>(dx);
  const fe W1p = fe_mul<C>(x1, Cp);
  const fe W2p = fe_mul<C>(x2, Cp);
  const fe dy = fe_sub<C>(y1, y2);
  const fe Dp = fe_sqr<C>(dy);
  const fe A1p = fe_mul<C>(y1, fe_sub<C>(W1p, W2p));
  const fe X3pc = fe_sub<C>(fe_sub<C>(Dp, W1p), W2p);
  const fe u = fe_sub<C>(X3pc, W1p);
  const fe Cc = fe_sqr<C>(u);
  // Z3 = Z * ((dx + X3' - W1')^2 - C' - C): as soon as C exists -- dx and C' die here instead of living across W1, W2 and A1
  if constexpr (!NOZ && z_early) {
    fe zz = fe_sqr<C>(fe_add<C>(dx, u));
    zz = fe_sub<C>(fe_sub<C>(zz, Cp), Cc);
    if constexpr (ZE == 2) z = fe_mul<C>(zz, z); else z = fe_mul<C>(z, zz);      // ZE == 2: the same product with the factors exchanged
  }
  const fe A1p2 = fe_dbl<C>(A1p);
  // Y3' = (dy + (W1' - X3'))^2 - D' - C - 2A1'
  fe yp = fe_sqr<C>(fe_sub<C>(dy, u));
  yp = fe_sub<C>(fe_sub<C>(yp, Dp), Cc);                 // = Y3' + 2A1'
  const fe Y3p = fe_sub<C>(yp, A1p2);
  // W1 = 4*X3'*C and W2 = 4*W1'*C: quadruple C once instead of X3' and W1' (same residues)
  const fe C4 = fe_shl<C, 2>(Cc);
  const fe W1 = fe_mul<C>(X3pc, C4);
  const fe W2 = fe_mul<C>(W1p, C4);
  fe ym = fe_sub<C>(Y3p, A1p2);
  // A1 = Y3' * (W1 - W2) is only ever subtracted from a product (y = ym * (W1 - x) - A1, twice): it stays an unreduced 512-bit
  // product and each y takes ONE reduction of the difference (fe_mul_sub_product) -- except with the reference's squaring, where
  // nothing about a squaring-free value changes but the instance is kept formula-for-formula.
  const fe2 A1wide = mul8x8(Y3p, fe_sub<C>(W1, W2));
  const fe W12 = fe_add<C>(W1, W2);
  // Z3 = Z * ((dx + X3' - W1')^2 - C' - C)
  if constexpr (!NOZ && !z_early) {
    fe zz = fe_sqr<C>(fe_add<C>(dx, u));
    zz = fe_sub<C>(fe_sub<C>(zz, Cp), Cc);
    z = fe_mul<C>(z, zz);
  }
  fe_cswap(oswap, ym, yp);
  const fe D = fe_sqr<C>(ym);
  x1 = fe_sub<C>(D, W12);
  y1 = fe_mul_sub_product<C>(ym, fe_sub<C>(W1, x1), A1wide);
  const fe Dc = fe_sqr<C>(yp);
  x2 = fe_sub<C>(Dc, W12);
  y2 = fe_mul_sub_product<C>(yp, fe_sub<C>(W1, x2), A1wide);
}

// curve_group.h:155-179 ADD_Z2_1: (X1,Y1,Z1) + affine (x2, y2) [Z2 = R mod p].  7M + 4S.
template <int C> ECS_DEV jpoint add_z2_1(const fe& X1, const fe& Y1, const fe& Z1, const fe& x2, const fe& y2) {
  const fe Z1Z1 = fe_sqr<C>(Z1);
  const fe U2 = fe_mul<C>(x2, Z1Z1);
  const fe S2 = fe_mul<C>(fe_mul<C>(y2, Z1), Z1Z1);
  const fe H = fe_sub<C>(U2, X1);
  const fe HH = fe_sqr<C>(H);
  const fe I = fe_shl<C, 2>(HH);
  const fe J = fe_mul<C>(H, I);
  const fe r = fe_dbl<C>(fe_sub<C>(S2, Y1));
  const fe V = fe_mul<C>(X1, I);
  jpoint R;
  R.x = fe_sub<C>(fe_sub<C>(fe_sqr<C>(r), J), fe_dbl<C>(V));
  R.y = fe_mul_sub_product<C>(r, fe_sub<C>(V, R.x), mul8x8(fe_dbl<C>(Y1), J));     // r*(V - X3) - 2*Y1*J: one reduction for the difference
  R.z = fe_sub<C>(fe_sub<C>(fe_sqr<C>(fe_add<C>(Z1, H)), Z1Z1), HH);
  return R;
}

// Mixed addition with 8M + 3S and only 7 linear operations (Hankerson-Menezes-Vanstone, Alg. 3.22) for the
// windowed fixed-base kernels: there the Jacobian representative is free (affine-level parity), and the
// linear operations -- canonical, ~20 VALU instructions each -- are what ADD_Z2_1's 15 of them cost.
// Z3 = Z1 * H (ADD_Z2_1 returns 2 * Z1 * H): a different representative of the same point.
template <int C> ECS_DEV jpoint madd_hmv(const fe& X1, const fe& Y1, const fe& Z1, const fe& x2, const fe& y2) {
  const fe Z1Z1 = fe_sqr<C>(Z1);
  const fe U2 = fe_mul<C>(x2, Z1Z1);
  const fe S2 = fe_mul<C>(y2, fe_mul<C>(Z1Z1, Z1));
  const fe H = fe_sub<C>(U2, X1);
  const fe r = fe_sub<C>(S2, Y1);
  const fe HH = fe_sqr<C>(H);
  const fe HHH = fe_mul<C>(H, HH);
  const fe V = fe_mul<C>(X1, HH);
  jpoint R;
  R.z = fe_mul<C>(Z1, H);
  R.x = fe_sub<C>(fe_sub<C>(fe_sqr<C>(r), HHH), fe_dbl<C>(V));
  R.y = fe_mul_sub_product<C>(r, fe_sub<C>(V, R.x), mul8x8(Y1, HHH));               // one reduction for the difference of two products
  return R;
}

// group order n (curve_nist_p256.h has no order: the reference never reduces scalars; SEC 2 values)
template <int CURVE> struct curve_order;
template <> struct curve_order<CURVE_P256> {
  static constexpr uint32_t N[8] = {0xfc632551u, 0xf3b9cac2u, 0xa7179e84u, 0xbce6faadu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0xffffffffu};
};
template <> struct curve_order<CURVE_SECP256K1> {
  static constexpr uint32_t N[8] = {0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
};

// a^(p-2) and a^((p+1)/4): exponents as compile-time word arrays (gfp.h:79-87).
template <int C> struct curve_exps;
template <> struct curve_exps<CURVE_P256> {
  static constexpr uint32_t P_M2[8]   = {0xfffffffdu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xffffffffu};
  static constexpr uint32_t P_SQRT[8] = {0x00000000u, 0x00000000u, 0x40000000u, 0x00000000u, 0x00000000u, 0x40000000u, 0xc0000000u, 0x3fffffffu};
};
template <> struct curve_exps<CURVE_SECP256K1> {
  static constexpr uint32_t P_M2[8]   = {0xfffffc2du, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  static constexpr uint32_t P_SQRT[8] = {0xbfffff0cu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
};

template <> struct curve_exps<CURVE_SECP256K1_CLASSICAL> : curve_exps<CURVE_SECP256K1> {};
template <> struct curve_exps<CURVE_P256_REFSQR> : curve_exps<CURVE_P256> {};
template <> struct curve_exps<CURVE_SECP256K1_REFSQR> : curve_exps<CURVE_SECP256K1> {};

// secp256k1: p - 2 = [223 ones][0][22 ones][0000][1][0][11][0][1];  (p + 1)/4 = [223 ones][0][22 ones][0000][11][00]
template <int C, bool SQRT> ECS_DEV fe secp256k1_pow_chain(const fe& x);
// a^(2^n): n successive squarings (a loop, not unrolled: the chains below run up to 128 of them)
template <int C> ECS_DEV fe fe_sqr_n(fe a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = fe_sqr<C>(a);
  return a;
}
#ifndef ECS_INVERSE_DIVSTEPS
#define ECS_INVERSE_DIVSTEPS 1       // 0: the a^(p-2) addition chains below (round 1)
#endif
// ---------------------------------------------------------------- modular inversion by divsteps ("safegcd")
// a^-1 mod p through the Bernstein-Yang division steps (CHES 2019; the constant-time, 30-bits-at-a-time formulation that
// libsecp256k1 documents as modinv32) instead of a^(p-2): 20 rounds of 30 division steps on the low words of (f, g) = (p, a) --
// 30 x 14 cheap 32-bit operations -- each followed by one 2x2-matrix update of the 9 x 30-bit signed limbs of (f, g) and of
// the Bezout pair (d, e) mod p (90 multiply-adds): ~12 000 instructions against ~50 000 for the 255 S + 12 M addition chain.
// The inverse of a residue is unique, so the result is the reference's a^(p-2) bit for bit (0 -> 0 included); control flow
// and addresses do not depend on a.  The reference-square instances keep the reference's own power ladder (fe_inverse).
template <int C> struct safegcd_consts;
template <> struct safegcd_consts<CURVE_P256> {
  static constexpr int32_t P30[9] = {0x3fffffff, 0x3fffffff, 0x3fffffff, 0x0000003f, 0x00000000, 0x00000000, 0x00001000, 0x3fffc000, 0x0000ffff};
  static constexpr uint32_t PINV30 = 0x3fffffffu;        // p^-1 mod 2^30
  static constexpr uint32_t R3[8] = {0x0000000au, 0xfffffffdu, 0xfffffff7u, 0xffffffedu, 0xfffffffcu, 0x00000005u, 0x00000001u, 0x00000018u};   // R^3 mod p
};
template <> struct safegcd_consts<CURVE_SECP256K1_CLASSICAL> {
  static constexpr int32_t P30[9] = {0x3ffffc2f, 0x3ffffffb, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x0000ffff};
  static constexpr uint32_t PINV30 = 0x2ddacacfu;
};

template <int C> ECS_DEV fe fe_inverse_divsteps(const fe& a) {
  using K = safegcd_consts<C>;
  constexpr int32_t M30 = 0x3fffffff;
  int32_t d[9], e[9], f[9], g[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { d[i] = 0; e[i] = 0; f[i] = K::P30[i]; }
  e[0] = 1;
  // g = a in 30-bit limbs
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 30 * i, j = bit >> 5, sh = bit & 31;
    uint32_t v = a.w[j] >> sh;
    if (sh > 2 && j + 1 < 8) v |= a.w[j + 1] << (32 - sh);
    g[i] = (int32_t)(v & (uint32_t)M30);
  }
  int32_t zeta = -1;
#pragma unroll 1
  for (int round = 0; round < 20; ++round) {
    // 30 division steps on the low words; (u, v; q, r) * (f, g) = 2^30 * (f', g')
    int32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t fl = (uint32_t)f[0] | ((uint32_t)f[1] << 30), gl = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
#pragma unroll
    for (int i = 0; i < 30; ++i) {
      int32_t c1 = zeta >> 31;
      const int32_t c2 = -(int32_t)(gl & 1u);
      const uint32_t x = (fl ^ (uint32_t)c1) - (uint32_t)c1;
      const int32_t y = (u ^ c1) - c1, z = (v ^ c1) - c1;
      gl += x & (uint32_t)c2; q += y & c2; r += z & c2;
      c1 &= c2;
      zeta = (zeta ^ c1) - 1;
      fl += gl & (uint32_t)c1; u += q & c1; v += r & c1;
      gl >>= 1; u <<= 1; v <<= 1;
    }
    // (d, e) <- (u d + v e, q d + r e) / 2^30 mod p: a multiple of p makes the low 30 bits vanish
    {
      const int32_t sd = d[8] >> 31, se = e[8] >> 31;
      int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
      int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0], ce = (int64_t)q * d[0] + (int64_t)r * e[0];
      md -= (int32_t)((K::PINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
      me -= (int32_t)((K::PINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
      cd += (int64_t)K::P30[0] * md; ce += (int64_t)K::P30[0] * me;
      cd >>= 30; ce >>= 30;
#pragma unroll
      for (int i = 1; i < 9; ++i) {
        cd += (int64_t)u * d[i] + (int64_t)v * e[i]; ce += (int64_t)q * d[i] + (int64_t)r * e[i];
        if (K::P30[i] != 0) { cd += (int64_t)K::P30[i] * md; ce += (int64_t)K::P30[i] * me; }
        d[i - 1] = (int32_t)cd & M30; e[i - 1] = (int32_t)ce & M30;
        cd >>= 30; ce >>= 30;
      }
      d[8] = (int32_t)cd; e[8] = (int32_t)ce;
    }
    // (f, g) <- (u f + v g, q f + r g) / 2^30 (exact)
    {
      int64_t cf = (int64_t)u * f[0] + (int64_t)v * g[0], cg = (int64_t)q * f[0] + (int64_t)r * g[0];
      cf >>= 30; cg >>= 30;
#pragma unroll
      for (int i = 1; i < 9; ++i) {
        cf += (int64_t)u * f[i] + (int64_t)v * g[i]; cg += (int64_t)q * f[i] + (int64_t)r * g[i];
        f[i - 1] = (int32_t)cf & M30; g[i - 1] = (int32_t)cg & M30;
        cf >>= 30; cg >>= 30;
      }
      f[8] = (int32_t)cf; g[8] = (int32_t)cg;
    }
  }
  // g = 0 and f = +-1 now: the inverse is sign(f) * d, brought into [0, p)
  {
    int32_t cond = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] += K::P30[i] & cond;
    const int32_t neg = f[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = (d[i] ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    cond = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] += K::P30[i] & cond;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
  }
  fe r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int bit = 32 * j, k = bit / 30, o = bit % 30;
    uint32_t w = (uint32_t)d[k] >> o;
    w |= (uint32_t)d[k + 1] << (30 - o);
    if (o > 28 && k + 2 < 9) w |= (uint32_t)d[k + 2] << (60 - o);
    r.w[j] = w;
  }
  if constexpr (C == CURVE_P256) return fe_mul<C>(r, fe_const<C, K::R3>());   // (a R)^-1 = a^-1 R^-1 (plain) -> a^-1 R: Montgomery-multiply by R^3
  else return r;
}

// gfp.h:42-44 inverse() = a^(p-2) and gfp.h:46-54 sqrt() = a^((p+1)/4).  The reference raises to the power with
// square-and-multiply over the exponent's bits (mgry_ops.h:44-86: 255 S + 128 M for P-256, 255 S + 249 M for
// secp256k1); the power of a canonical residue does not depend on how the exponent is walked, so fixed addition
// chains over the runs of ones give the same bits with 255 S + 12 M (P-256), 255 S + 15 M and 253 S + 13 M
// (secp256k1, the chain libsecp256k1 documents).  Exponents checked against p - 2 and (p + 1)/4 symbolically.
template <int C> ECS_DEV fe fe_inverse(const fe& x) {
  if constexpr (curve_prime<C>::ref_square) {
    // reference-compatible squaring: the squarings the reference performs, in its order (mgry_ops.h:44-86)
    return fe_pow<C>(x, curve_exps<C>::P_M2);
#if ECS_INVERSE_DIVSTEPS
  } else if constexpr (C == CURVE_P256 || C == CURVE_SECP256K1_CLASSICAL) {
    return fe_inverse_divsteps<C>(x);
#endif
  } else if constexpr (C == CURVE_P256) {
    // p - 2 = [32 ones][31 zeros][1][96 zeros][94 ones][0][1]
    const fe x2 = fe_mul<C>(fe_sqr<C>(x), x);
    const fe x3 = fe_mul<C>(fe_sqr<C>(x2), x);
    const fe x6 = fe_mul<C>(fe_sqr_n<C>(x3, 3), x3);
    const fe x12 = fe_mul<C>(fe_sqr_n<C>(x6, 6), x6);
    const fe x15 = fe_mul<C>(fe_sqr_n<C>(x12, 3), x3);
    const fe x30 = fe_mul<C>(fe_sqr_n<C>(x15, 15), x15);
    const fe x32 = fe_mul<C>(fe_sqr_n<C>(x30, 2), x2);
    fe t = fe_mul<C>(fe_sqr_n<C>(x32, 32), x);
    t = fe_mul<C>(fe_sqr_n<C>(t, 128), x32);
    t = fe_mul<C>(fe_sqr_n<C>(t, 32), x32);
    t = fe_mul<C>(fe_sqr_n<C>(t, 30), x30);
    return fe_mul<C>(fe_sqr_n<C>(t, 2), x);
  } else {
    return secp256k1_pow_chain<C, false>(x);
  }
}
template <int C, bool SQRT> ECS_DEV fe secp256k1_pow_chain(const fe& x) {
  const fe x2 = fe_mul<C>(fe_sqr<C>(x), x);
  const fe x3 = fe_mul<C>(fe_sqr<C>(x2), x);
  const fe x6 = fe_mul<C>(fe_sqr_n<C>(x3, 3), x3);
  const fe x9 = fe_mul<C>(fe_sqr_n<C>(x6, 3), x3);
  const fe x11 = fe_mul<C>(fe_sqr_n<C>(x9, 2), x2);
  const fe x22 = fe_mul<C>(fe_sqr_n<C>(x11, 11), x11);
  const fe x44 = fe_mul<C>(fe_sqr_n<C>(x22, 22), x22);
  const fe x88 = fe_mul<C>(fe_sqr_n<C>(x44, 44), x44);
  const fe x176 = fe_mul<C>(fe_sqr_n<C>(x88, 88), x88);
  const fe x220 = fe_mul<C>(fe_sqr_n<C>(x176, 44), x44);
  const fe x223 = fe_mul<C>(fe_sqr_n<C>(x220, 3), x3);
  fe t = fe_mul<C>(fe_sqr_n<C>(x223, 23), x22);
  if constexpr (SQRT) {
    t = fe_mul<C>(fe_sqr_n<C>(t, 6), x2);
    return fe_sqr_n<C>(t, 2);
  } else {
    t = fe_mul<C>(fe_sqr_n<C>(t, 5), x);
    t = fe_mul<C>(fe_sqr_n<C>(t, 3), x2);
    return fe_mul<C>(fe_sqr_n<C>(t, 2), x);
  }
}
// The same two chains on fe29.cuh's 29-bit limbs (round 4, ECS_SQRT_RADIX): 253 squarings in a row are what the carry-free columns are best at; every
// operand is a product (tight), the value is the canonical one at the end (leave29), so the bits are the chains' above.
#ifndef ECS_SQRT_RADIX
#define ECS_SQRT_RADIX 29
#endif
template <int C> ECS_DEV fe29 sqr29_n(fe29 a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = sqr29<C>(a);
  return a;
}
template <int C> ECS_DEV fe fe_sqrt_candidate29(const fe& xin) {
  const fe29 x = enter29<C>(xin);
  const fe29 x2 = mul29<C>(sqr29<C>(x), x);
  if constexpr (C == CURVE_P256) {
    const fe29 x4 = mul29<C>(sqr29_n<C>(x2, 2), x2);
    const fe29 x8 = mul29<C>(sqr29_n<C>(x4, 4), x4);
    const fe29 x16 = mul29<C>(sqr29_n<C>(x8, 8), x8);
    const fe29 x32 = mul29<C>(sqr29_n<C>(x16, 16), x16);
    fe29 t = mul29<C>(sqr29_n<C>(x32, 32), x);
    t = mul29<C>(sqr29_n<C>(t, 96), x);
    return leave29<C>(sqr29_n<C>(t, 94));
  } else {
    const fe29 x3 = mul29<C>(sqr29<C>(x2), x);
    const fe29 x6 = mul29<C>(sqr29_n<C>(x3, 3), x3);
    const fe29 x9 = mul29<C>(sqr29_n<C>(x6, 3), x3);
    const fe29 x11 = mul29<C>(sqr29_n<C>(x9, 2), x2);
    const fe29 x22 = mul29<C>(sqr29_n<C>(x11, 11), x11);
    const fe29 x44 = mul29<C>(sqr29_n<C>(x22, 22), x22);
    const fe29 x88 = mul29<C>(sqr29_n<C>(x44, 44), x44);
    const fe29 x176 = mul29<C>(sqr29_n<C>(x88, 88), x88);
    const fe29 x220 = mul29<C>(sqr29_n<C>(x176, 44), x44);
    const fe29 x223 = mul29<C>(sqr29_n<C>(x220, 3), x3);
    fe29 t = mul29<C>(sqr29_n<C>(x223, 23), x22);
    t = mul29<C>(sqr29_n<C>(t, 6), x2);
    return leave29<C>(sqr29_n<C>(t, 2));
  }
}
template <int C> ECS_DEV fe fe_sqrt_candidate(const fe& x) {        // a^((p+1)/4): a square root if there is one (p = 3 mod 4)
  if constexpr (curve_prime<C>::ref_square) {
    return fe_pow<C>(x, curve_exps<C>::P_SQRT);            // the reference's own sequence of squarings (mgry_ops.h:44-86)
  } else if constexpr (ECS_SQRT_RADIX == 29 && (C == CURVE_P256 || C == CURVE_SECP256K1_CLASSICAL)) {
    return fe_sqrt_candidate29<C>(x);
  } else if constexpr (C == CURVE_P256) {
    // (p + 1)/4 = (2^32 - 1) 2^222 + 2^190 + 2^94: 253 S + 7 M (bit by bit it is 253 S + 33 M; the power does not depend on the walk)
    const fe x2 = fe_mul<C>(fe_sqr<C>(x), x);
    const fe x4 = fe_mul<C>(fe_sqr_n<C>(x2, 2), x2);
    const fe x8 = fe_mul<C>(fe_sqr_n<C>(x4, 4), x4);
    const fe x16 = fe_mul<C>(fe_sqr_n<C>(x8, 8), x8);
    const fe x32 = fe_mul<C>(fe_sqr_n<C>(x16, 16), x16);
    fe t = fe_mul<C>(fe_sqr_n<C>(x32, 32), x);
    t = fe_mul<C>(fe_sqr_n<C>(t, 96), x);
    return fe_sqr_n<C>(t, 94);
  } else {
    return secp256k1_pow_chain<C, true>(x);
  }
}

// API domain (Montgomery form) <-> the domain the multiplication-heavy code runs in
// (curve_domain<C>::fast: Montgomery for P-256, classical for secp256k1 -- field.cuh).
template <int C> ECS_DEV fe to_fast(const fe& v) {
  if constexpr (curve_domain<C>::fast == C) return v; else return fe_to_classical<C>(v);
}
template <int C> ECS_DEV fe from_fast(const fe& v) {
  if constexpr (curve_domain<C>::fast == C) return v; else return fe_from_classical<C>(v);
}
// fast domain -> classical
template <int C> ECS_DEV fe fast_to_classical(const fe& v) {
  if constexpr (curve_domain<C>::fast == C) return fe_to_classical<C>(v); else return v;
}
// classical -> fast domain
template <int C> ECS_DEV fe classical_to_fast(const fe& v) {
  if constexpr (curve_domain<C>::fast == C) return fe_from_classical<C>(v); else return v;
}

// jacobian_curve_point.h:33-42 to_affine: one inversion per lane; returns classical (x, y).
template <int C> ECS_DEV void to_affine(const jpoint& P, fe& ax, fe& ay) {     // P in the API's Montgomery form
  constexpr int CI = curve_domain<C>::fast;
  const fe invZ = fe_inverse<CI>(to_fast<C>(P.z));
  const fe invZ2 = fe_sqr<CI>(invZ);
  const fe invZ3 = fe_mul<CI>(invZ2, invZ);
  ax = fast_to_classical<C>(fe_mul<CI>(to_fast<C>(P.x), invZ2));
  ay = fast_to_classical<C>(fe_mul<CI>(to_fast<C>(P.y), invZ3));
}

// curve_group.h:189-218 scalar_mult: co-Z Joye double-add ladder, LSB -> MSB, fixed 254 ZDAU
// iterations, k forced odd and corrected at the end with ADD_Z2_1(P, -P0).  (xm, ym) is the base
// point in Montgomery form with implicit Z = R mod p.  `kwords` points at this lane's 8 scalar
// words in global memory: one word is (re)read per 32 iterations so the scalar does not occupy
// VGPRs across the ZDAU body.
// The ladder up to (not including) the even-k correction: (px, py) = k'P with k' = k | 1, (bx, by) the other register, common z.
template <int C, bool NOZ> ECS_DEV uint32_t ladder_core(const uint32_t* __restrict__ kwords, const fe& xm, const fe& ym, fe& px, fe& py, fe& bx, fe& by, fe& z) {
  px = xm; py = ym;
  // base = TRPLU(P): DBLU then ZADDU (curve_group.h:183-186)
  {
    fe dx2, dy2;
    dblu<C>(px, py, dx2, dy2, z);
    zaddu<C>(px, py, dx2, dy2, z, bx, by);
  }
  uint32_t kw = kwords[0];
  const uint32_t k0 = kw;
  // The reference runs swap(m_1), then swap(m_b); ZDAU; swap(m_b) for b = 2..255 (curve_group.h:196-211; m_b =
  // utility.h:45-51 wide_mask_bit of bit b).  Two swaps in a row compose into one by the XOR of their masks, and the
  // swap after a ZDAU is folded into the ZDAU's own output (zdau's `oswap`): one field-element swap per bit.
  uint32_t cur = 0u - ((kw >> 2) & 1u);                  // m_2
  {
    const uint32_t m = (0u - ((kw >> 1) & 1u)) ^ cur;    // swap(m_1) then swap(m_2)
    fe_cswap(m, px, bx);                                 // swap.h:47-56 swap_if_same_z
    fe_cswap(m, py, by);
  }
#pragma unroll 1
  for (int b = 2; b < 256; ++b) {
    const int nb = b + 1;
    if ((nb & 31) == 0) kw = (nb < 256) ? kwords[nb >> 5] : 0u;       // one word per 32 bits: the scalar is not kept in VGPRs
    const uint32_t next = 0u - ((kw >> (nb & 31)) & 1u);              // m_(b+1); 0 after the last bit: the closing swap(m_255)
    zdau<C, NOZ>(bx, by, px, py, z, cur ^ next);                      // base = ZDAU(base, P), outputs swapped by m_b ^ m_(b+1)
    cur = next;
  }
  return k0;
}

// The same ladder with its 254 ZDAU iterations on fe29.cuh's reduced-radix representation (round 4): TRPLU and the opening swaps as
// above on canonical words, then the co-Z pair and Z enter the radix-2^29 Montgomery domain (one product by a constant each), the loop
// runs zdau29, and (px, py, z) = k'P leave it as the canonical residues the 8-word loop would have produced -- the field VALUES of
// every iteration are the same, so the result is bit-identical (the representation of an intermediate is nobody's business).
// Which instances have it: every one whose squaring is exact (the reference-square twins depend on the 32-bit Montgomery digits).
template <int C> struct ladder_has_radix29 { static constexpr bool value = (C == CURVE_P256 || C == CURVE_SECP256K1_CLASSICAL); };
template <int C, bool NOZ = false> ECS_DEV uint32_t ladder_core29(const uint32_t* __restrict__ kwords, const fe& xm, const fe& ym, fe& px, fe& py, fe& bx, fe& by, fe& z) {
  px = xm; py = ym;
  {
    fe dx2, dy2;
    dblu<C>(px, py, dx2, dy2, z);
    zaddu<C>(px, py, dx2, dy2, z, bx, by);
  }
  uint32_t kw = kwords[0];
  const uint32_t k0 = kw;
  uint32_t cur = 0u - ((kw >> 2) & 1u);
  {
    const uint32_t m = (0u - ((kw >> 1) & 1u)) ^ cur;
    fe_cswap(m, px, bx);
    fe_cswap(m, py, by);
  }
  coz29 s;
  s.x1 = enter29<C>(bx); s.x2 = enter29<C>(px); s.y1 = enter29<C>(by); s.z = enter29<C>(z);
  s.dx = sub29(s.x1, s.x2);
  s.dy = sub29(s.y1, enter29<C>(py));
#pragma unroll 1
  for (int b = 2; b < 256; ++b) {
    const int nb = b + 1;
    if ((nb & 31) == 0) kw = (nb < 256) ? kwords[nb >> 5] : 0u;
    const uint32_t next = 0u - ((kw >> (nb & 31)) & 1u);
    zdau29<C, NOZ>(s, cur ^ next);
    cur = next;
  }
  px = leave29<C>(s.x2); py = leave29<C>(sub29(s.y1, s.dy));
  if constexpr (NOZ) { bx = leave29<C>(s.x1); by = leave29<C>(s.y1); }       // the x-only ladder reads both registers; Z was not carried
  else z = leave29<C>(s.z);
  return k0;
}

template <int C, int RADIX = 32> ECS_DEV jpoint scalar_mult_ladder(const uint32_t* __restrict__ kwords, const fe& xm, const fe& ym) {
  fe px, py, bx, by, z;
  uint32_t k0;
  if constexpr (RADIX == 29) k0 = ladder_core29<C>(kwords, xm, ym, px, py, bx, by, z);
  else k0 = ladder_core<C, false>(kwords, xm, ym, px, py, bx, by, z);
  // even k: subtract the original point once (curve_group.h:214-217)
  const fe oppy = fe_opposite<C>(ym);                    // jacobian_curve_point.h:48-54 via gfp.h:60-64
  const jpoint Psub = add_z2_1<C>(px, py, z, xm, oppy);
  const uint32_t meven = 0u - (uint32_t)((k0 & 1u) == 0u);
  jpoint R;
  R.x = fe_select(meven, Psub.x, px);
  R.y = fe_select(meven, Psub.y, py);
  R.z = fe_select(meven, Psub.z, z);
  return R;
}

// x(k*P) for ODD k* without carrying Z (a != 0 curves: P-256).  Not the reference's return value -- an x-coordinate-only
// product for callers like ECDH: the loop is the reference's ladder minus the Z update (8M + 6S per bit instead of 9M + 7S).
// At the end the two registers (X0, Y0) = k*P and (X1, Y1) share an unknown Z; with w = Z^2 the curve equation
// Y^2 = X^3 + a X w^2 + b w^3 holds for both, so
//     D = (Y0^2 - X0^3) - (Y1^2 - X1^3) = a (X0 - X1) w^2        G = X1 (Y0^2 - X0^3) - X0 (Y1^2 - X1^3) = b (X1 - X0) w^3
// give w = -(a / b) G / D and the affine x = X0 / w = X0 * b * D / (-a * G) = num / den (one simultaneous inversion over the
// batch follows).  Only Z^2 is determined by this -- the sign of Z, hence y, is not (both are consistent with everything but the
// input point), which is why the reference-identical Jacobian result cannot be had this way.
// With a = 0 (secp256k1) only w^3 is determined and x keeps a cube-root ambiguity: not offered there.
#ifndef ECS_LADDER_X_RADIX
#define ECS_LADDER_X_RADIX 29
#endif
template <int C> ECS_DEV void scalar_mult_ladder_x(const uint32_t* __restrict__ kwords_odd, const fe& xm, const fe& ym, fe& num, fe& den) {
  static_assert(curve_prime<C>::is_p256, "a = -3");
  fe x0, y0, x1, y1, z;
#if ECS_LADDER_X_RADIX == 29
  (void)ladder_core29<C, true>(kwords_odd, xm, ym, x0, y0, x1, y1, z);      // round 4: the 254 iterations on 29-bit limbs, like the reference ladder's
#else
  (void)ladder_core<C, true>(kwords_odd, xm, ym, x0, y0, x1, y1, z);
#endif
  const fe e0 = fe_sub<C>(fe_sqr<C>(y0), fe_mul<C>(fe_sqr<C>(x0), x0));
  const fe e1 = fe_sub<C>(fe_sqr<C>(y1), fe_mul<C>(fe_sqr<C>(x1), x1));
  const fe D = fe_sub<C>(e0, e1);
  const fe G = fe_mul_sub_product<C>(x1, e0, mul8x8(x0, e1));
  num = fe_mul<C>(fe_mul<C>(x0, FE_CONST(C, BM)), D);
  den = fe_add<C>(fe_dbl<C>(G), G);                      // -a * G = 3G
}

}  // namespace ecsimd_hip
