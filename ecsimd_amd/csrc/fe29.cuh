// fe29.cuh -- reduced-radix prime-field arithmetic for the ladder's 254-iteration loop (round 4): nine SIGNED 29-bit limbs.
//
// Why a second representation.  field.cuh keeps a field element in 8 x 32-bit words and returns canonical residues; per ladder
// iteration that is 828 multiplies but 2 033 carry-chain additions (v_addc_co_u32: 4.3 issue cycles each, as dear as a multiply),
// 40 conditional subtractions and 236 column moves -- 12 580 cycles of which the multiplies are 3 600.
// Here a field element is value = sum l[i] * 2^(29 i) with nine signed 32-bit limbs and the Montgomery radix R' = 2^261:
//   * a column of a product is at most 9 x 2^58 x (small factors) < 2^63, so v_mad_i64_i32 accumulates a whole column in its own
//     64-bit accumulator -- NO carry instruction between the products of a column;
//   * additions and subtractions are nine v_add_u32 / v_sub_u32 (2.3 cycles each) with no carry chain and no conditional
//     subtraction: limbs and values are allowed to grow within proven bounds ("lazy"), and one parallel carry pass (norm29)
//     brings an operand back where a square needs it;
//   * the Montgomery reduction is folded into the column walk (finely integrated product scanning): q_k is the column's low
//     29 bits (p = -1 mod 2^29 for P-256), and q_k * p enters four later columns through p's sparse signed form
//     2^256 - 2^224 + 2^192 + 2^96 - 1 -- 36 multiply-adds by constants per reduction; a column ends with one v_and_b32 and one
//     v_ashrrev_i64.  secp256k1 (2^256 - 2^32 - 977): q_k = column * 977^-1 mod 2^29, three constants.
// The VALUES computed are the reference's (curve_group.h:120-153): x -> x * 2^261 is a field isomorphism like x -> x * 2^256, the
// kernel converts at the loop's boundary and canonicalises at the end, so the ladder's X, Y, Z stay bit-identical (level J).
//
// The bounds that make this sound are PROVEN, not assumed: tools/radix29_model.py executes these same functions (same names, same
// order) on intervals and shows that one zdau29 maps the loop invariant into itself with every limb inside int32 and every
// column inside int64 (tests/test_radix29_model.py), and on concrete integers against the big-int ZDAU.
#pragma once
#include <utility>
#include "field.cuh"

namespace ecsimd_hip {

constexpr int R29_LIMBS = 9;
constexpr int R29_BITS = 29;
constexpr int32_t R29_MASK = (1 << R29_BITS) - 1;

struct fe29 { int32_t l[R29_LIMBS]; };

// A uniform constant the optimiser cannot see through: keeps `q * 2^9` a v_mad_i64_i32 by an SGPR instead of a 64-bit shift + add.
ECS_DEV int32_t r29_opaque(int32_t c) { int32_t r; asm("s_mov_b32 %0, %1" : "=s"(r) : "i"(c)); return r; }
// x + x as ONE dual-rate v_add_u32 (the compiler's canonical form is a shift)
ECS_DEV int32_t r29_dbl32(int32_t x) { int32_t r; asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x)); return r; }

template <int C> struct r29_prime;                 // the prime's sparse signed form in radix 2^29 (tools/radix29_model.py Curve.terms)
// p256: P-256's reduction and a = -3 formulas; dense: any odd p < 2^256 as nine tight limbs in SGPRs (round 5: curves registered at run time);
// tight_sq: zdau29 carry-passes dy - u and dx + u before squaring them (the interval proof needs it for every reduction but secp256k1's sparse one)
template <> struct r29_prime<CURVE_P256> { static constexpr bool p256 = true, dense = false, tight_sq = true; };
template <> struct r29_prime<CURVE_SECP256K1_CLASSICAL> { static constexpr bool p256 = false, dense = false, tight_sq = false; static constexpr uint32_t QMUL = 0x12253531u; };   // 977^-1 mod 2^29
template <> struct r29_prime<CURVE_GENERIC> { static constexpr bool p256 = false, dense = true, tight_sq = true; };
// What the column walk needs beside its operands: nothing for the built-in primes (an empty object, a defaulted last argument of every function below),
// the prime itself for a registered curve -- wave-uniform values that arrive as a kernel argument and live in SGPRs (tools/radix29_model.py Curve.dense).
template <int C> struct r29_ctx {};
template <> struct r29_ctx<CURVE_GENERIC> {
  int32_t p[9];        // p in tight limbs: limbs 0..7 in [0, 2^29), the top one < 2^24
  uint32_t qinv;       // -p^-1 mod 2^29: q_k = column * qinv makes column + q_k p vanish mod 2^29
  int32_t in[9];       // 2^266 mod p, tight: API Montgomery form x 2^256 -> x 2^261
  int32_t out[9];      // 2^256 mod p, tight: back
};

ECS_DEV fe29 add29(const fe29& a, const fe29& b) { fe29 r;
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) r.l[i] = a.l[i] + b.l[i];
  return r; }
ECS_DEV fe29 sub29(const fe29& a, const fe29& b) { fe29 r;
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) r.l[i] = a.l[i] - b.l[i];
  return r; }
ECS_DEV fe29 dbl29(const fe29& a) { fe29 r;
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) r.l[i] = r29_dbl32(a.l[i]);
  return r; }
// One PARALLEL carry pass over (a << SHIFT): every carry is taken from the input limbs, so limbs 0..7 end in [c_min, 2^29 + c_max)
// with |c| <= 4 -- "normalised", which is what a square's operand has to be; the top limb keeps the value's sign and excess.
template <int SHIFT = 0> ECS_DEV fe29 norm29(const fe29& a) {
  fe29 r;
  auto shl = [](int32_t v) { return (int32_t)((uint32_t)v << SHIFT); };
  r.l[0] = shl(a.l[0]) & R29_MASK;
#pragma unroll
  for (int i = 1; i < R29_LIMBS - 1; ++i) r.l[i] = (shl(a.l[i]) & R29_MASK) + (a.l[i - 1] >> (R29_BITS - SHIFT));
  r.l[R29_LIMBS - 1] = shl(a.l[R29_LIMBS - 1]) + (a.l[R29_LIMBS - 2] >> (R29_BITS - SHIFT));
  return r;
}
ECS_DEV void cswap29(uint32_t m, fe29& a, fe29& b) {
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) { const int32_t t = (a.l[i] ^ b.l[i]) & (int32_t)m; a.l[i] ^= t; b.l[i] ^= t; }
}

#include "fe29_cols.inc"

// Montgomery product a * b / 2^261 (SQR: a * a / 2^261 with the 36 cross products taken once on a doubled operand), reduction
// folded into the column walk.  Result: limbs 0..7 in [0, 2^29), the top limb signed and small; value in (T / R', T / R' + p).
// Column K is ONE asm statement (fe29_cols.inc, generated): its products and the quotient digits' contributions accumulate on top
// of the carry; the compiler only extracts the digit (v_and_b32) and shifts the column down (v_ashrrev_i64) between two statements.
struct r29_consts_sgpr { int32_t k9, k18, km21, k24, km8, km977; };
template <int C, bool SQR, int K> ECS_DEV void fips29_column(int64_t& acc, const fe29& a, const int32_t (&a2)[R29_LIMBS], const fe29& b,
                                                             int32_t (&q)[R29_LIMBS], fe29& r, const r29_consts_sgpr& k, const r29_ctx<C>& cx) {
  using PR = r29_prime<C>;
  constexpr int lo = K > R29_LIMBS - 1 ? K - (R29_LIMBS - 1) : 0, hi = K < R29_LIMBS - 1 ? K : R29_LIMBS - 1;
  constexpr int NCROSS = SQR ? ((K - 1) / 2 >= lo && K > 0 ? (K - 1) / 2 - lo + 1 : 0) : hi - lo + 1;
  constexpr int NP = NCROSS + ((SQR && K % 2 == 0) ? 1 : 0);
  int32_t x[NP > 0 ? NP : 1], y[NP > 0 ? NP : 1];
#pragma unroll
  for (int t = 0; t < NCROSS; ++t) { x[t] = SQR ? a2[lo + t] : a.l[lo + t]; y[t] = SQR ? a.l[K - lo - t] : b.l[K - lo - t]; }
  if constexpr (SQR && K % 2 == 0) { x[NCROSS] = a.l[K / 2]; y[NCROSS] = a.l[K / 2]; }
  constexpr auto in = [](int j) { return j >= 0 && j < R29_LIMBS; };
  if constexpr (PR::p256) {
    // + q 2^9 three limbs up, + q 2^18 six up, - q 2^21 seven up, + q 2^24 eight up; - q at its own column cancels the low 29 bits
    constexpr int NR = (in(K - 3) ? 1 : 0) + (in(K - 6) ? 1 : 0) + (in(K - 7) ? 1 : 0) + (in(K - 8) ? 1 : 0);
    int32_t qq[NR > 0 ? NR : 1], cc[NR > 0 ? NR : 1];
    int n = 0;
    if constexpr (in(K - 3)) { qq[n] = q[K - 3]; cc[n++] = k.k9; }
    if constexpr (in(K - 6)) { qq[n] = q[K - 6]; cc[n++] = k.k18; }
    if constexpr (in(K - 7)) { qq[n] = q[K - 7]; cc[n++] = k.km21; }
    if constexpr (in(K - 8)) { qq[n] = q[K - 8]; cc[n++] = k.k24; }
    if constexpr (K == 0) r29_col_first<NP>::run(acc, x, y); else r29_col<NP, NR>::run(acc, x, y, qq, cc);
    if constexpr (K < R29_LIMBS) q[K] = (int32_t)acc & R29_MASK;          // acc - q[K] is a multiple of 2^29: the shift below drops it
    else r.l[K - R29_LIMBS] = (int32_t)acc & R29_MASK;
  } else if constexpr (PR::dense) {
    // any odd p: + q_j p_(K-j) for every quotient digit already known (j < K), then q_K = column * (-p^-1) mod 2^29 and + q_K p_0 clears the low 29 bits
    constexpr int jlo = K > R29_LIMBS - 1 ? K - (R29_LIMBS - 1) : 0, jhi = K - 1 < R29_LIMBS - 1 ? K - 1 : R29_LIMBS - 1;
    constexpr int NR = jhi >= jlo ? jhi - jlo + 1 : 0;
    int32_t qq[NR > 0 ? NR : 1], cc[NR > 0 ? NR : 1];
#pragma unroll
    for (int t = 0; t < NR; ++t) { qq[t] = q[jlo + t]; cc[t] = cx.p[K - jlo - t]; }
    if constexpr (K == 0) r29_col_first<NP>::run(acc, x, y); else r29_col<NP, NR>::run(acc, x, y, qq, cc);
    if constexpr (K < R29_LIMBS) { q[K] = (int32_t)((uint32_t)acc * cx.qinv) & R29_MASK; acc += (int64_t)q[K] * cx.p[0]; }
    else r.l[K - R29_LIMBS] = (int32_t)acc & R29_MASK;
  } else {
    // secp256k1: - 8 q one limb up, + q 2^24 eight up; - 977 q at its own column, after q = column * 977^-1 mod 2^29
    constexpr int NR = (in(K - 1) ? 1 : 0) + (in(K - 8) ? 1 : 0);
    int32_t qq[NR > 0 ? NR : 1], cc[NR > 0 ? NR : 1];
    int n = 0;
    if constexpr (in(K - 1)) { qq[n] = q[K - 1]; cc[n++] = k.km8; }
    if constexpr (in(K - 8)) { qq[n] = q[K - 8]; cc[n++] = k.k24; }
    if constexpr (K == 0) r29_col_first<NP>::run(acc, x, y); else r29_col<NP, NR>::run(acc, x, y, qq, cc);
    if constexpr (K < R29_LIMBS) { q[K] = (int32_t)((uint32_t)acc * PR::QMUL) & R29_MASK; acc += (int64_t)q[K] * k.km977; }
    else r.l[K - R29_LIMBS] = (int32_t)acc & R29_MASK;
  }
  acc >>= R29_BITS;
}
template <int C, bool SQR, int... K> ECS_DEV void fips29_columns(int64_t& acc, const fe29& a, const int32_t (&a2)[R29_LIMBS], const fe29& b,
                                                                int32_t (&q)[R29_LIMBS], fe29& r, const r29_consts_sgpr& k, const r29_ctx<C>& cx, std::integer_sequence<int, K...>) {
  (fips29_column<C, SQR, K>(acc, a, a2, b, q, r, k, cx), ...);
}
template <int C, bool SQR> ECS_DEV fe29 fips29(const fe29& a, const fe29& b, const r29_ctx<C>& cx) {
  int32_t q[R29_LIMBS];
  int32_t a2[R29_LIMBS];
  fe29 r;
  if constexpr (SQR) {
#pragma unroll
    for (int i = 0; i < R29_LIMBS; ++i) a2[i] = r29_dbl32(a.l[i]);
  }
  const r29_consts_sgpr k{r29_opaque(1 << 9), r29_opaque(1 << 18), r29_opaque(-(1 << 21)), r29_opaque(1 << 24), r29_opaque(-8), r29_opaque(-977)};
  int64_t acc;
  fips29_columns<C, SQR>(acc, a, a2, b, q, r, k, cx, std::make_integer_sequence<int, 2 * R29_LIMBS - 1>{});
  r.l[R29_LIMBS - 1] = (int32_t)acc;
  return r;
}
template <int C> ECS_DEV fe29 mul29(const fe29& a, const fe29& b, const r29_ctx<C>& cx = r29_ctx<C>{}) { return fips29<C, false>(a, b, cx); }
template <int C> ECS_DEV fe29 sqr29(const fe29& a, const r29_ctx<C>& cx = r29_ctx<C>{}) { return fips29<C, true>(a, a, cx); }

// ---------------------------------------------------------------- conversions at the loop's boundary
// canonical 8 x 32-bit words (a value < 2^256) -> nine tight limbs of the same integer
ECS_DEV fe29 to29(const fe& v) {
  fe29 r;
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) {
    const int bit = R29_BITS * i, j = bit >> 5, sh = bit & 31;
    uint32_t w = v.w[j] >> sh;
    if (sh > 32 - R29_BITS && j + 1 < 8) w |= v.w[j + 1] << (32 - sh);
    r.l[i] = (int32_t)(w & (uint32_t)R29_MASK);
  }
  return r;
}
// nine tight limbs of an integer < 2^256 -> 8 words
ECS_DEV fe from29(const fe29& t) {
  fe r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int bit = 32 * j, k = bit / R29_BITS, o = bit % R29_BITS;
    uint32_t w = (uint32_t)t.l[k] >> o;
    w |= (uint32_t)t.l[k + 1] << (R29_BITS - o);
    if (2 * R29_BITS - o < 32 && k + 2 < R29_LIMBS) w |= (uint32_t)t.l[k + 2] << (2 * R29_BITS - o);
    r.w[j] = w;
  }
  return r;
}
template <int C> struct r29_consts;
template <> struct r29_consts<CURVE_P256> {
  // tight limbs (tools/radix29_model.py to_limbs) of p, of 2^266 mod p (field.cuh's x * 2^256 -> x * 2^261) and of 2^256 mod p (back)
  static constexpr int32_t P[9]   = {0x1fffffff, 0x1fffffff, 0x1fffffff, 0x000001ff, 0x00000000, 0x00000000, 0x00040000, 0x1fe00000, 0x00ffffff};
  static constexpr int32_t IN[9]  = {0x00000400, 0x00000000, 0x00000000, 0x1ff80000, 0x1fffffff, 0x1fffffff, 0x0fffffff, 0x1fffffff, 0x00000003};
  static constexpr int32_t OUT[9] = {0x00000001, 0x00000000, 0x00000000, 0x1ffffe00, 0x1fffffff, 0x1fffffff, 0x1ffbffff, 0x001fffff, 0x00000000};
  static constexpr int32_t ONE[9] = {0x00000020, 0x00000000, 0x00000000, 0x1fffc000, 0x1fffffff, 0x1fffffff, 0x1f7fffff, 0x03ffffff, 0x00000000};   // 2^261 mod p: the field's 1
};
template <> struct r29_consts<CURVE_SECP256K1_CLASSICAL> {
  // the secp256k1 loops run in the CLASSICAL domain of field.cuh: in = 2^522 mod p (x -> x * 2^261), out = 1
  static constexpr int32_t P[9]   = {0x1ffffc2f, 0x1ffffff7, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x00ffffff};
  static constexpr int32_t IN[9]  = {0x1a428400, 0x00f44001, 0x00010000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000};
  static constexpr int32_t OUT[9] = {0x00000001, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000};
  static constexpr int32_t ONE[9] = {0x00007a20, 0x00000100, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000};   // 2^261 mod p = 32 (2^32 + 977)
};
template <const int32_t (&ARR)[9]> ECS_DEV fe29 fe29_const() { fe29 r;
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) r.l[i] = ARR[i];
  return r; }
ECS_DEV fe29 fe29_from(const int32_t (&arr)[9]) { fe29 r;
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) r.l[i] = arr[i];
  return r; }
// the constants of the loop's boundary: compile-time tables for the built-in primes, the context's for a registered curve
template <int C> ECS_DEV fe29 r29_in(const r29_ctx<C>& cx) { if constexpr (r29_prime<C>::dense) return fe29_from(cx.in); else return fe29_const<r29_consts<C>::IN>(); }
template <int C> ECS_DEV fe29 r29_out(const r29_ctx<C>& cx) { if constexpr (r29_prime<C>::dense) return fe29_from(cx.out); else return fe29_const<r29_consts<C>::OUT>(); }
template <int C> ECS_DEV int32_t r29_p(const r29_ctx<C>& cx, int i) { if constexpr (r29_prime<C>::dense) return cx.p[i]; else return r29_consts<C>::P[i]; }
// a field element of the loop's surroundings (canonical, field.cuh's fast domain) -> x * 2^261 in tight limbs
template <int C> ECS_DEV fe29 enter29(const fe& v, const r29_ctx<C>& cx = r29_ctx<C>{}) { return mul29<C>(to29(v), r29_in<C>(cx), cx); }
// ... and back: the canonical residue of field.cuh's domain.  `v` is anything the loop holds (|value| < 8 p): the product with a
// tight constant < p lies in (-p/4, 9p/8), + p makes it positive, a sequential carry pass makes the limbs tight, and two
// conditional subtractions of p (sign of the top limb after a borrow pass) land in [0, p).
template <int C> ECS_DEV fe canon29(fe29 t, const r29_ctx<C>& cx = r29_ctx<C>{}) {       // a value in (-p, 2p) (leave29 needs (-p, 9p/8)): + p, carry pass, two conditional subtractions
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) t.l[i] += r29_p<C>(cx, i);
#pragma unroll
  for (int i = 0; i < R29_LIMBS - 1; ++i) { t.l[i + 1] += t.l[i] >> R29_BITS; t.l[i] &= R29_MASK; }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    fe29 d;
#pragma unroll
    for (int i = 0; i < R29_LIMBS; ++i) d.l[i] = t.l[i] - r29_p<C>(cx, i);
#pragma unroll
    for (int i = 0; i < R29_LIMBS - 1; ++i) { d.l[i + 1] += d.l[i] >> R29_BITS; d.l[i] &= R29_MASK; }
    const int32_t keep = d.l[R29_LIMBS - 1] >> 31;                     // all ones where t < p
#pragma unroll
    for (int i = 0; i < R29_LIMBS; ++i) t.l[i] = (t.l[i] & keep) | (d.l[i] & ~keep);
  }
  return from29(t);
}
template <int C> ECS_DEV fe leave29(const fe29& v, const r29_ctx<C>& cx = r29_ctx<C>{}) { return canon29<C>(mul29<C>(v, r29_out<C>(cx), cx), cx); }

// ---------------------------------------------------------------- the ladder iteration
// Loop state: the co-Z pair (x1, y1), (x2, y2) with y2 kept as dy = y1 - y2 and dx = x1 - x2 carried beside x1, x2 -- differences of
// TIGHT values, so the next iteration squares them without a carry pass (tools/radix29_model.py ladder_invariant has the intervals).
struct coz29 { fe29 x1, x2, dx, y1, dy, z; };

// One iteration: (x1, y1) <- 2 (x1, y1) + (x2, y2), (x2, y2) re-expressed with the new z; `oswap` exchanges the two outputs.
// The field VALUES are point.cuh zdau<C>'s (curve_group.h:120-153), statement for statement in tools/radix29_model.py zdau29.
// Differences from the 8-word form: the factor 4 of W1 = 4 X3' C, W2 = 4 W1' C rides in on a normalised 4C (one pass makes both
// products tight and true-valued); A1 is an ordinary product (a shared 18-column product is dearer than the reduction it saves here).
// NOZ: the x-only ladder's iteration (point.cuh scalar_mult_ladder_x) -- the same without the Z update, 8M + 6S.
template <int C, bool NOZ = false> ECS_DEV void zdau29(coz29& s, uint32_t oswap, const r29_ctx<C>& cx = r29_ctx<C>{}) {
  const fe29 Cp = sqr29<C>(s.dx, cx);
  const fe29 W1p = mul29<C>(s.x1, Cp, cx);
  const fe29 W2p = mul29<C>(s.x2, Cp, cx);
  const fe29 Dp = sqr29<C>(s.dy, cx);
  const fe29 A1p = mul29<C>(s.y1, sub29(W1p, W2p), cx);
  const fe29 X3 = sub29(sub29(Dp, W1p), W2p);
  const fe29 u = norm29(sub29(X3, W1p));
  const fe29 Cc = sqr29<C>(u, cx);
  // (two of the carry passes are not secp256k1's: with its sparse reduction the interval proof holds with dy - u and dx + u squared as they are)
  constexpr bool TIGHT_SQ = r29_prime<C>::tight_sq;
  const fe29 dyu = sub29(s.dy, u);
  fe29 yp = norm29(sub29(sub29(sqr29<C>(TIGHT_SQ ? norm29(dyu) : dyu, cx), Dp), Cc));      // Y3' + 2 A1'
  const fe29 A2 = dbl29(A1p);
  const fe29 Y3p = sub29(yp, A2);
  fe29 ym = norm29(sub29(Y3p, A2));
  const fe29 C4 = norm29<2>(Cc);
  const fe29 W1 = mul29<C>(X3, C4, cx);
  const fe29 W2 = mul29<C>(W1p, C4, cx);
  const fe29 A1 = mul29<C>(Y3p, sub29(W1, W2), cx);
  if constexpr (!NOZ) {
    const fe29 dxu = add29(s.dx, u);
    const fe29 zz = sub29(sub29(sqr29<C>(TIGHT_SQ ? norm29(dxu) : dxu, cx), Cp), Cc);
    s.z = mul29<C>(s.z, zz, cx);
  }
  cswap29(oswap, ym, yp);
  const fe29 D = sqr29<C>(ym, cx);
  const fe29 Dc = sqr29<C>(yp, cx);
  const fe29 W12 = add29(W1, W2);
  s.x1 = sub29(D, W12);
  s.x2 = sub29(Dc, W12);
  s.dx = sub29(D, Dc);
  const fe29 P1 = mul29<C>(ym, sub29(W1, s.x1), cx);
  const fe29 P2 = mul29<C>(yp, sub29(W1, s.x2), cx);
  s.y1 = sub29(P1, A1);
  s.dy = sub29(P1, P2);
}


// ---------------------------------------------------------------- the combs' mixed addition (round 4)
// A Jacobian accumulator (X, Y, Z) + an affine table point (x2, y2), Hankerson-Menezes-Vanstone Alg. 3.22 as point.cuh madd_hmv<C> (8M + 3S),
// on the reduced-radix representation.  Invariant of the accumulator (tools/radix29_model.py comb_invariant, proven by prove_comb_invariant:
// one madd29 maps it into itself with every limb inside int32 and every column inside int64): X limbs in [-3, 1] x 2^29, Y in [-1, 1] x 2^29,
// Z tight; table coordinates tight, y possibly negated.  Three carry passes: H and r feed squares, V - X3 is 31 bits wide before its product.
struct jpoint29 { fe29 x, y, z; };
// (Two halves, so that a caller can look at H and r -- R = +-T is H = 0, then r = 0: is_zero29 below -- before paying for the rest.)
template <int C> ECS_DEV void madd29_hr(const jpoint29& P, const fe29& x2, const fe29& y2, fe29& H, fe29& r, const r29_ctx<C>& cx = r29_ctx<C>{}) {
  const fe29 Z1Z1 = sqr29<C>(P.z, cx);
  const fe29 U2 = mul29<C>(x2, Z1Z1, cx);
  const fe29 S2 = mul29<C>(y2, mul29<C>(Z1Z1, P.z, cx), cx);
  H = norm29(sub29(U2, P.x));
  r = norm29(sub29(S2, P.y));
}
template <int C> ECS_DEV jpoint29 madd29_finish(const jpoint29& P, const fe29& H, const fe29& r, const r29_ctx<C>& cx = r29_ctx<C>{}) {
  const fe29 HH = sqr29<C>(H, cx);
  const fe29 HHH = mul29<C>(H, HH, cx);
  const fe29 V = mul29<C>(P.x, HH, cx);
  jpoint29 R;
  R.z = mul29<C>(P.z, H, cx);
  R.x = sub29(sub29(sqr29<C>(r, cx), HHH), dbl29(V));
  R.y = sub29(mul29<C>(r, norm29(sub29(V, R.x)), cx), mul29<C>(P.y, HHH, cx));
  return R;
}
template <int C> ECS_DEV jpoint29 madd29(const jpoint29& P, const fe29& x2, const fe29& y2, const r29_ctx<C>& cx = r29_ctx<C>{}) {
  fe29 H, r; madd29_hr<C>(P, x2, y2, H, r, cx); return madd29_finish<C>(P, H, r, cx); }
// ---------------------------------------------------------------- the variable-base window loop (round 4)
// Doublings multiply by 3, 4 and 8, and on lazy limbs nothing ever takes a multiple of p away: a Montgomery product only divides by 2^261 ~ 32 p,
// so values above ~10 p GROW from one doubling to the next.  vred29 is the missing piece: v -> v - k p with k = round(top limb / 2^24) -- the top
// limb IS the value in units of 2^232, p is 2^24 of them -- through p's sparse signed form: five (secp256k1: three) full-rate limb updates, after
// which |v| < p/2 + a few 2^232.  Applied to X3 and Y3 of every doubling and of the double-add, it closes the loop's invariant
// (tools/radix29_model.py prove_window_invariant: limbs in [-2.25, 1.25] x 2^29, |value| <= 0.6 p, Z tight).
// (A dense prime -- a registered curve -- has no sparse form to subtract through, and needs none: its window loop, gjdbl29 below, forms 8 Y^4 as
// 2 (2 YY)^2, not as 8 x a product, so no value is ever multiplied by more than 4 after its last division by 2^261 and the loop's values close
// on their own; tools/radix29_model.py prove_gwindow_invariant.  vred29 is the identity there.)
template <int C> ECS_DEV fe29 vred29(fe29 a) {
  if constexpr (r29_prime<C>::dense) return a;
  const int32_t k = (a.l[8] + (1 << 23)) >> 24;
  if constexpr (r29_prime<C>::p256) { a.l[0] += k; a.l[3] -= k << 9; a.l[6] -= k << 18; a.l[7] += k << 21; }
  else { a.l[0] += 977 * k; a.l[1] += k << 3; }
  a.l[8] -= k << 24;
  return a;
}
// Jacobian doubling (a = -3: 4M + 4S, a = 0: 3M + 5S... the formulas of k_varwin.inc jdbl): YY = Y^2, G = 4 YY, B = X G, alpha = 3 (X - Z^2)(X + Z^2) | 3 X^2,
// X3 = alpha^2 - 2 B, Y3 = alpha (B - X3) - 8 YY^2, Z3 = 2 Y Z.  (No halving here: 8 Y^4 is formed as 2 x (4 YY^2) by a shifting carry pass.)
template <int C> ECS_DEV jpoint29 jdbl29(const jpoint29& P) {
  const fe29 Yn = norm29(P.y);
  const fe29 YY = sqr29<C>(Yn);
  const fe29 G = norm29<2>(YY);
  const fe29 B = mul29<C>(P.x, G);
  fe29 t;
  if constexpr (r29_prime<C>::p256) {
    const fe29 delta = sqr29<C>(P.z);
    t = mul29<C>(sub29(P.x, delta), norm29(add29(P.x, delta)));          // (one carry pass is enough: the interval proof holds with X - delta as it is)
  } else {
    t = sqr29<C>(norm29(P.x));
  }
  const fe29 alpha = norm29(add29(dbl29(t), t));
  jpoint29 R;
  R.z = mul29<C>(dbl29(Yn), P.z);
  const fe29 X3 = sub29(sqr29<C>(alpha), dbl29(B));
  const fe29 E8 = dbl29(norm29<2>(sqr29<C>(YY)));
  R.y = vred29<C>(sub29(mul29<C>(alpha, sub29(B, X3)), E8));              // (B - X3 needs no carry pass beside the carry-passed alpha)
  R.x = vred29<C>(X3);
  return R;
}
// 2R + T for an affine T = (x2, y2) as (R + T) + R: the mixed addition's by-products X1 H^2, Y1 H^3 are R in the coordinates of R + T, so the second
// addition is a co-Z one (k_varwin.inc dbl_add: 13M + 5S).
template <int C> ECS_DEV jpoint29 dbl_add29(const jpoint29& P, const fe29& x2, const fe29& y2, const r29_ctx<C>& cx = r29_ctx<C>{}) {
  const fe29 Z1Z1 = sqr29<C>(P.z, cx);
  const fe29 U2 = mul29<C>(x2, Z1Z1, cx);
  const fe29 S2 = mul29<C>(y2, mul29<C>(Z1Z1, P.z, cx), cx);
  const fe29 H = norm29(sub29(U2, P.x));
  const fe29 r = norm29(sub29(S2, P.y));
  const fe29 HH = sqr29<C>(H, cx);
  const fe29 HHH = mul29<C>(H, HH, cx);
  const fe29 V = mul29<C>(P.x, HH, cx);
  const fe29 Yh = mul29<C>(P.y, HHH, cx);
  const fe29 X3 = sub29(sub29(sqr29<C>(r, cx), HHH), dbl29(V));
  const fe29 Y3 = sub29(mul29<C>(r, norm29(sub29(V, X3)), cx), Yh);
  const fe29 Z3 = mul29<C>(P.z, H, cx);
  const fe29 dx = norm29(sub29(X3, V));
  const fe29 dy = norm29(sub29(Y3, Yh));
  const fe29 Cc = sqr29<C>(dx, cx);
  const fe29 W1 = mul29<C>(X3, Cc, cx);
  const fe29 W2 = mul29<C>(V, Cc, cx);
  const fe29 A1 = mul29<C>(Y3, sub29(W1, W2), cx);
  jpoint29 Q;
  const fe29 Qx = sub29(sub29(sqr29<C>(dy, cx), W1), W2);
  Q.y = vred29<C>(sub29(mul29<C>(dy, sub29(W1, Qx), cx), A1));           // (nor W1 - Qx beside the carry-passed dy)
  Q.x = vred29<C>(Qx);
  Q.z = mul29<C>(Z3, dx, cx);
  return Q;
}
// The doubling of a curve with ANY coefficient a (round 5: the window loop of a curve registered at run time, k_gvarwin.hip), in modified Jacobian
// coordinates (Cohen-Miyaji-Ono): w = a Z^4 rides beside the point, so alpha = 3 X^2 + w costs no product, and the doubled point's w is 2 (8 Y^4) w --
// one product (WOUT; the last doubling before an addition does not need it).  3M + 4S (+ 1M), against 4M + 6S with a Z^4 formed from Z every time.
// 8 Y^4 is 2 (2 YY)^2: a SQUARE of a carry-passed double -- never 8 x a product, whose + p of the reduction would come back as + 8 p, the growth that
// vred29 is there to undo where the prime has a sparse form.  With it the values close on their own (|X|, |Y| < 3.3 x 2^256 from one doubling
// to the next): no value reduction on a dense prime.  Five carry passes (tools/radix29_model.py gjdbl29, prove_gwindow_invariant: any odd p < 2^256).
template <int C, bool WOUT> ECS_DEV jpoint29 gjdbl29(const jpoint29& P, fe29& w, const r29_ctx<C>& cx = r29_ctx<C>{}) {
  const fe29 Yn = norm29(P.y);
  const fe29 YY = sqr29<C>(Yn, cx);
  const fe29 G = norm29<2>(YY);
  const fe29 B = mul29<C>(P.x, G, cx);
  const fe29 XX = sqr29<C>(norm29(P.x), cx);
  const fe29 alpha = norm29(add29(add29(dbl29(XX), XX), w));
  jpoint29 R;
  R.z = mul29<C>(dbl29(Yn), P.z, cx);
  const fe29 X3 = sub29(sqr29<C>(alpha, cx), dbl29(B));
  const fe29 E4 = sqr29<C>(norm29<1>(YY), cx);
  R.y = sub29(mul29<C>(alpha, norm29(sub29(B, X3)), cx), dbl29(E4));
  R.x = X3;
  if constexpr (WOUT) { w = mul29<C>(norm29<2>(E4), w, cx); }
  return R;
}
// The mixed addition BETWEEN DOUBLINGS (the default GLV loop, k_varwin.inc k_varwin_mult_glv): X3 takes one more carry pass, X3 and Y3 the value
// reduction -- then the sum lies inside the window loop's invariant again (tools/radix29_model.py prove_glv_invariant).
template <int C> ECS_DEV jpoint29 madd29v_finish(const jpoint29& P, const fe29& H, const fe29& r) {
  jpoint29 R = madd29_finish<C>(P, H, r);
  R.x = vred29<C>(norm29(R.x)); R.y = vred29<C>(R.y);
  return R;
}
// The co-Z addition with update (curve_group.h:91-116 ZADDU, 5M + 2S): (x1, y1) + (x2, y2) over their common z -> (rx, ry); (x1, y1) is re-expressed
// over the new z = z dx; dx = x1 - x2 (carry-passed) is handed out -- the ratio of the two Z (k_varwin.inc k_varwin_table_iso walks back through them).
// (x1, y1) tight products, (x2, y2) a jdbl29 output or a sum of this function: tools/radix29_model.py iso_chain_invariant, prove_iso_table.
template <int C> ECS_DEV void zaddu29(fe29& x1, fe29& y1, const fe29& x2, const fe29& y2, fe29& z, fe29& rx, fe29& ry, fe29& dx, const r29_ctx<C>& cx = r29_ctx<C>{}) {
  dx = norm29(sub29(x1, x2));
  const fe29 Cc = sqr29<C>(dx, cx);
  const fe29 W1 = mul29<C>(x1, Cc, cx);
  const fe29 W2 = mul29<C>(x2, Cc, cx);
  const fe29 dy = norm29(sub29(y1, y2));
  const fe29 D = sqr29<C>(dy, cx);
  const fe29 A1 = mul29<C>(y1, sub29(W1, W2), cx);
  rx = sub29(sub29(D, W1), W2);
  ry = sub29(mul29<C>(dy, sub29(W1, rx), cx), A1);
  z = mul29<C>(z, dx, cx);
  x1 = W1; y1 = A1;
}
// v = 0 as a FIELD element, for |value| < 2^260: the value reduction leaves |v| < p, where the only multiple of p is the integer 0; a sequential
// carry pass makes the limbs below the top one canonical, so the integer 0 is nine zero limbs.  Its caller branches on the result: public scalars only.
template <int C> ECS_DEV bool is_zero29(fe29 v) {
  v = vred29<C>(v);
#pragma unroll
  for (int i = 0; i < R29_LIMBS - 1; ++i) { v.l[i + 1] += v.l[i] >> R29_BITS; v.l[i] &= R29_MASK; }
  int32_t d = 0;
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) d |= v.l[i];
  return d == 0;
}

// ---------------------------------------------------------------- the complete addition law of a = 0 curves (round 4; k_varwin.inc k_varwin_mult_glv_ct)
// Homogeneous projective (X : Y : Z), Renes-Costello-Batina 2016 as k_varwin.inc pdbl_complete / padd_mixed_complete, for b = 7 (3b = 21).
// 21 x = 4x + 16x + x by two shifting carry passes, then vred29 and a carry pass (the result feeds products and triplings).  Carry passes
// where the interval proof needs them (tools/radix29_model.py prove_complete_invariant: every coordinate with limbs in [-2.25, 3.25] x 2^29 and
// |value| <= 1.05 p between two operations), vred29 on every output.
template <int C> ECS_DEV fe29 mul21_29(const fe29& x) {
  const fe29 a = norm29<2>(x);
  const fe29 b = norm29<2>(a);
  return vred29<C>(add29(add29(a, b), x));
}
template <int C> ECS_DEV jpoint29 pdbl29(const jpoint29& P) {
  const fe29 Yn = norm29(P.y), Zn = norm29(P.z);
  const fe29 yy = sqr29<C>(Yn), zz = sqr29<C>(Zn);
  const fe29 xy = mul29<C>(P.x, Yn), yz = mul29<C>(Yn, Zn);
  const fe29 t = norm29(mul21_29<C>(zz));
  const fe29 m = sub29(yy, add29(dbl29(t), t));
  const fe29 q = norm29(add29(yy, t));
  jpoint29 R;
  R.x = vred29<C>(dbl29(mul29<C>(xy, m)));
  R.y = vred29<C>(add29(mul29<C>(m, q), dbl29(norm29<2>(mul29<C>(yy, t)))));
  R.z = vred29<C>(dbl29(norm29<2>(mul29<C>(yy, yz))));
  return R;
}
template <int C> ECS_DEV jpoint29 padd29(const jpoint29& P, const fe29& x2, const fe29& y2) {
  const fe29 Xn = norm29(P.x), Yn = norm29(P.y), Zn = norm29(P.z);
  const fe29 t0 = mul29<C>(Xn, x2), t1 = mul29<C>(Yn, y2);
  const fe29 t3 = sub29(sub29(mul29<C>(add29(Xn, Yn), norm29(add29(x2, y2))), t0), t1);
  const fe29 t4 = norm29(add29(mul29<C>(y2, Zn), Yn));
  const fe29 t5 = norm29(add29(mul29<C>(x2, Zn), Xn));
  const fe29 z3b = norm29(mul21_29<C>(Zn));
  const fe29 A = sub29(t1, z3b), B = add29(t1, z3b);
  const fe29 Cc = mul21_29<C>(t5);
  const fe29 t03 = norm29(add29(dbl29(t0), t0));
  jpoint29 R;
  R.x = vred29<C>(sub29(mul29<C>(t3, A), mul29<C>(Cc, t4)));
  R.y = vred29<C>(add29(mul29<C>(t03, Cc), mul29<C>(B, A)));
  R.z = vred29<C>(add29(mul29<C>(t4, B), mul29<C>(t03, t3)));
  return R;
}

// -a where m is all ones, a where it is zero: (a ^ m) - m, two full-rate instructions per limb
ECS_DEV fe29 cneg29(uint32_t m, const fe29& a) { fe29 r;
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) r.l[i] = (a.l[i] ^ (int32_t)m) - (int32_t)m;
  return r; }
ECS_DEV fe29 select29(uint32_t m, const fe29& a, const fe29& b) { fe29 r;                 // m ? a : b
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) r.l[i] = b.l[i] ^ ((a.l[i] ^ b.l[i]) & (int32_t)m);
  return r; }
// a table coordinate: the canonical fast-domain element v  ->  the canonical residue of v * 2^261 as 8 words; to29 gives tight limbs back at the read
template <int C> ECS_DEV fe pack29(const fe& v) { return canon29<C>(enter29<C>(v)); }     // enter29 returns a value in [0, 1.04 p): canonicalised, it fits 8 words

}  // namespace ecsimd_hip
