// gfield.cuh -- the reference's field layer for ANY odd 256-bit modulus given at RUN time, one element per lane.
//
// The reference's L3 layer is generic in the modulus type P: details::mgry_reduce<P> (mgry_mul.h:84-121), mgry_constants<WBN, P>
// (mgry_csts.h:15-35), mgry_mul_constants (mgry_mul.h:25-50), mod_add / mod_sub / mod_shift_left_one (modular.h:10-41),
// mgry_pow (mgry_ops.h:44-86), GFp<WBN, P> (gfp.h:17-115; p = 3 mod 4 only, gfp.h:84).  field.cuh serves the two curve primes with
// their special forms; this header serves every other modulus -- above all the two GROUP ORDERS n, which ECDSA needs -- from a
// `gmod` the host derives once per modulus (capi.hip modulus registry) and passes BY VALUE as a kernel argument: it is wave-uniform, so
// the compiler keeps it in SGPRs and no constant memory or table is involved.
//
// Every function returns the canonical residue, like the reference's (sub.h:46-69 ends every op), so results are bit-identical to
// the reference instantiated with the same P.  Nothing here is on the scalar-multiplication hot path: the code is plain C++ over
// field.cuh's carry-chain helpers, ~3x the instruction count of the special-form primes (a generic word-serial reduction).
#pragma once
#include "field.cuh"

namespace ecsimd_hip {

struct gmod {
  uint32_t p[8];        // the modulus, odd, >= 3 (little-endian words)
  uint32_t r[8];        // R mod p, R = 2^256                           mgry_csts.h:20
  uint32_t rsq[8];      // R^2 mod p                                    mgry_csts.h:21
  uint32_t negr[8];     // (p - 1) * R mod p = -R mod p                  mgry_csts.h:24
  uint32_t r3[8];       // R^3 mod p: Montgomery form of a plain inverse of a Montgomery-form value
  uint32_t pm2[8];      // p - 2                                        gfp.h:79-81
  uint32_t psqrt[8];    // (p + 1) / 4 (meaningful for p = 3 mod 4)      gfp.h:84-87
  int32_t p30[9];       // p in signed 30-bit limbs (division steps)
  uint32_t pinv30;      // p^-1 mod 2^30
  uint32_t mprime;      // -p^-1 mod 2^32                               mgry_mul.h:33-38
  uint32_t flags;       // GMOD_*
};
enum : uint32_t { GMOD_PRIME = 1u, GMOD_3MOD4 = 2u };

ECS_DEV fe g_words(const uint32_t (&w)[8]) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = w[i];
  return r;
}
// r = (r + top * 2^256 >= p) ? r - p : r                                    sub.h:46-69
ECS_DEV void g_cond_sub(fe& r, lane_mask top, const gmod& M) {
  fe d;
  const fe P = g_words(M.p);
  const uint32_t borrow = sub8_3(d, r, P);                    // all-ones where r < p
  lane_mask lt;
  asm("v_cmp_ne_u32_e64 %0, %1, 0" : "=s"(lt) : "v"(borrow));
  const lane_mask keep = lt & ~top;
#pragma unroll
  for (int i = 0; i < 8; ++i) asm("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r.w[i]) : "v"(d.w[i]), "s"(keep));
}
// (a + b) mod p                                                              modular.h:10-15
ECS_DEV fe g_add(const fe& a, const fe& b, const gmod& M) {
  fe s;
  const lane_mask c = add8m3(s, a, b);
  g_cond_sub(s, c, M);
  return s;
}
// (a - b) mod p: subtract, add p back where it borrowed                      modular.h:24-41
ECS_DEV fe g_sub(const fe& a, const fe& b, const gmod& M) {
  fe d;
  const uint32_t m = sub8_3(d, a, b);
  fe P = g_words(M.p);
#pragma unroll
  for (int i = 0; i < 8; ++i) P.w[i] &= m;
  fe r;
  (void)add8m3(r, d, P);
  return r;
}
// 2a mod p                                                                   modular.h:17-22
ECS_DEV fe g_dbl(const fe& a, const gmod& M) {
  fe s;
  lane_mask c;
  asm("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(c) : "v"(a.w[7]));
#pragma unroll
  for (int i = 7; i > 0; --i) s.w[i] = __builtin_amdgcn_alignbit(a.w[i], a.w[i - 1], 31);
  s.w[0] = a.w[0] << 1;
  g_cond_sub(s, c, M);
  return s;
}
// T * 2^-256 mod p: 8 rounds of q = t_i * m' mod 2^32, t += q * p * 2^(32 i), then one conditional subtraction   mgry_mul.h:84-121
ECS_DEV fe g_reduce(fe2& t, const gmod& M) {
  uint32_t top = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t q = t.w[i] * M.mprime;
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t acc = (uint64_t)t.w[i + j] + carry;            // q * p_j + t + carry < 2^64
      acc += (uint64_t)q * M.p[j];
      t.w[i + j] = (uint32_t)acc;
      carry = (uint32_t)(acc >> 32);
    }
    const uint64_t s = (uint64_t)t.w[i + 8] + carry + top;
    t.w[i + 8] = (uint32_t)s;
    top = (uint32_t)(s >> 32);
  }
  fe res;
#pragma unroll
  for (int i = 0; i < 8; ++i) res.w[i] = t.w[8 + i];
  g_cond_sub(res, mask_nonzero(top), M);
  return res;
}
ECS_DEV fe g_mul(const fe& a, const fe& b, const gmod& M) {                    // mgry_ops.h:31-35
  fe2 t = mul8x8(a, b);
  return g_reduce(t, M);
}
// REF: the reference's square() as written (mul.h:160-212, dropped carry included -- field.cuh sqr8_ref)
template <bool REF> ECS_DEV fe g_sqr(const fe& a, const gmod& M) {             // mgry_ops.h:37-42
  fe2 t;
  if constexpr (REF) t = sqr8_ref(a); else t = sqr8(a);
  return g_reduce(t, M);
}
ECS_DEV fe g_from_classical(const fe& n, const gmod& M) { return g_mul(n, g_words(M.rsq), M); }     // mgry.h:47-50
ECS_DEV fe g_to_classical(const fe& n, const gmod& M) {                                             // mgry.h:52-55
  fe2 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) { t.w[i] = n.w[i]; t.w[8 + i] = 0; }
  return g_reduce(t, M);
}
ECS_DEV fe g_opposite(const fe& a, const gmod& M) {                                                 // gfp.h:60-64
  return g_sub(g_words(M.negr), g_sub(a, g_words(M.r), M), M);
}
// a^e, e public and wave-uniform; the multiplication sequence of mgry_ops.h:44-86
template <bool REF> ECS_DEV fe g_pow(const fe& a, const uint32_t (&e)[8], const gmod& M) {
  fe result = g_words(M.r);
  int top = -1;
  for (int i = 255; i >= 0; --i) if ((e[i >> 5] >> (i & 31)) & 1u) { top = i; break; }
  fe base = a;
#pragma unroll 1
  for (int i = 0; i <= top; ++i) {
    if ((e[i >> 5] >> (i & 31)) & 1u) result = g_mul(result, base, M);
    if (i < top) base = g_sqr<REF>(base, M);
  }
  return result;
}

// The PLAIN inverse x * a = 1 (mod p) of a plain residue a, gcd(a, p) = 1, by Bernstein-Yang division steps (point.cuh
// fe_inverse_divsteps with the modulus in SGPRs): 20 rounds of 30 steps, constant control flow and addresses.  0 -> 0.
// For a PRIME p this is a^(p-2) of the plain value; a composite p has no such identity (the registry only routes prime moduli here).
ECS_DEV fe g_inverse_plain(const fe& a, const gmod& M) {
  constexpr int32_t M30 = 0x3fffffff;
  int32_t d[9], e[9], f[9], g[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { d[i] = 0; e[i] = 0; f[i] = M.p30[i]; }
  e[0] = 1;
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
    {
      const int32_t sd = d[8] >> 31, se = e[8] >> 31;
      int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
      int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0], ce = (int64_t)q * d[0] + (int64_t)r * e[0];
      md -= (int32_t)((M.pinv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
      me -= (int32_t)((M.pinv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
      cd += (int64_t)M.p30[0] * md; ce += (int64_t)M.p30[0] * me;
      cd >>= 30; ce >>= 30;
#pragma unroll
      for (int i = 1; i < 9; ++i) {
        cd += (int64_t)u * d[i] + (int64_t)v * e[i]; ce += (int64_t)q * d[i] + (int64_t)r * e[i];
        cd += (int64_t)M.p30[i] * md; ce += (int64_t)M.p30[i] * me;
        d[i - 1] = (int32_t)cd & M30; e[i - 1] = (int32_t)ce & M30;
        cd >>= 30; ce >>= 30;
      }
      d[8] = (int32_t)cd; e[8] = (int32_t)ce;
    }
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
  {
    int32_t cond = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] += M.p30[i] & cond;
    const int32_t neg = f[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = (d[i] ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    cond = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] += M.p30[i] & cond;
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
  return r;
}
// Montgomery form in and out: (aR)^-1 = a^-1 R^-1 (plain), times R^3 by one Montgomery multiplication -> a^-1 R        gfp.h:42-44 for prime p
ECS_DEV fe g_inverse_mgry(const fe& a, const gmod& M) { return g_mul(g_inverse_plain(a, M), g_words(M.r3), M); }

ECS_DEV bool g_is_zero(const fe& a) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d |= a.w[i];
  return d == 0;
}
// all ones where a == 0, as data (no compare, no branch)
ECS_DEV uint32_t g_zero_mask(const fe& a) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d |= a.w[i];
  return (uint32_t)((int32_t)((d | (0u - d)) ^ 0x80000000u) >> 31);
}
// a < b as 256-bit integers
ECS_DEV bool g_less(const fe& a, const fe& b) { fe d; return sub8_3(d, a, b) != 0; }

}  // namespace ecsimd_hip
