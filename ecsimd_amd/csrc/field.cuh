// field.cuh -- 256-bit prime-field arithmetic for gfx950 (CDNA4), one field element per lane.
//
// Replaces the reference's L2/L3 layers (include/ecsimd/{add,sub,mul,shift,modular,mgry_mul,
// mgry_ops}.h) for the two curves on the hot path.  A field element is 8 x u32 words in VGPRs
// (= the 4 x u64 little-endian limbs of the reference's bignum_256, bignum.h:97-99); every
// function returns the CANONICAL residue in [0, p), exactly like the reference
// (sub.h:46-69 sub_if_above ends every op), so any expression DAG evaluated with these
// functions is bit-identical to the reference's (SURVEY.md 8(a) "parity level J").
//
// gfx950 issue costs measured with tools/ubench/valu_rates.hip (cycles per wave64 instruction per
// SIMD at >= 2 waves/SIMD):  v_mad_u64_u32 4.4, v_addc/subb_co_u32 4.35, v_cndmask_b32 (SGPR mask)
// 4.1, v_alignbit 4.1, v_mov/v_and/v_xor/v_add_u32 2.2-2.4.  The compiler lowers a carry-checked
// 64-bit MAC to 4+ instructions (mad + lshl_add_u64 + cmp_lt_u64 + cndmask), so the carry chains
// below are written as inline asm: each statement is one self-contained chain (VCC defined and
// consumed inside the statement), the compiler only allocates registers and schedules statements.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ecsimd_hip {

// CURVE_SECP256K1_CLASSICAL is internal: the same field and curve as CURVE_SECP256K1 with elements
// kept in the CLASSICAL domain (x instead of x*R), where p = 2^256 - 2^32 - 977 allows a
// pseudo-Mersenne reduction.  x -> x*R is a field isomorphism, so a formula evaluated in either
// domain yields the same element; kernels convert at their boundary and the C ABI never sees it.
//
// CURVE_*_REFSQR (ECSIMD_HIP_REF_SQUARE_COMPAT): the same curves with the reference's square() as it is written,
// mul.h:160-212 -- including the carry it drops at mul.h:186-190 (its own "TODO: carry?", mul.h:207).  Opt-in, for
// callers that need the reference's bits on the ~3e-6 of scalar multiplications where square(a) != mul(a, a).
// The dropped carry depends on the Montgomery-form digits, so these instances stay in the Montgomery domain.
// CURVE_GENERIC (round 5): a curve registered at RUN time -- the reference's curve_group<Curve> takes any Curve type (curve.h:12-15); its constants travel as a
// kernel argument (gcurve.cuh), the field layer is gfield.cuh's, the ladder's loop fe29.cuh's with the dense 9-limb prime in SGPRs (r29_ctx<CURVE_GENERIC>).
enum : int { CURVE_P256 = 0, CURVE_SECP256K1 = 1, CURVE_SECP256K1_CLASSICAL = 2, CURVE_P256_REFSQR = 3, CURVE_SECP256K1_REFSQR = 4, CURVE_GENERIC = 5 };

struct fe { uint32_t w[8]; };                  // little-endian 32-bit words
struct fe2 { uint32_t w[16]; };                // 512-bit product

#define ECS_DEV __device__ __forceinline__

// ---------------------------------------------------------------- per-curve constants
// Values pinned against the reference in tests/test_constants.py.
template <int CURVE> struct curve_consts;

template <> struct curve_consts<CURVE_P256> {
  // p = 2^256 - 2^224 + 2^192 + 2^96 - 1                       curve_nist_p256.h:17-19
  static constexpr uint32_t P[8]    = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xffffffffu};
  static constexpr uint32_t R_P[8]  = {0x00000001u, 0x00000000u, 0x00000000u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xfffffffeu, 0x00000000u};   // R mod p    mgry_csts.h:20
  static constexpr uint32_t RSQ[8]  = {0x00000003u, 0x00000000u, 0xffffffffu, 0xfffffffbu, 0xfffffffeu, 0xffffffffu, 0xfffffffdu, 0x00000004u};   // R^2 mod p  mgry_csts.h:21
  static constexpr uint32_t NEG_R[8] = {0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x00000001u, 0x00000000u, 0x00000000u, 0x00000002u, 0xfffffffeu};  // (p-1)*R mod p  mgry_csts.h:24
  static constexpr uint32_t AM[8]   = {0xfffffffcu, 0xffffffffu, 0xffffffffu, 0x00000003u, 0x00000000u, 0x00000000u, 0x00000004u, 0xfffffffcu};   // a*R mod p  curve_group.h:32
  static constexpr uint32_t BM[8]   = {0x29c4bddfu, 0xd89cdf62u, 0x78843090u, 0xacf005cdu, 0xf7212ed6u, 0xe5a220abu, 0x04874834u, 0xdc30061du};   // b*R mod p  curve_group.h:31
  static constexpr uint32_t GX[8]   = {0xd898c296u, 0xf4a13945u, 0x2deb33a0u, 0x77037d81u, 0x63a440f2u, 0xf8bce6e5u, 0xe12c4247u, 0x6b17d1f2u};   // curve_nist_p256.h:27-29
  static constexpr uint32_t GY[8]   = {0x37bf51f5u, 0xcbb64068u, 0x6b315eceu, 0x2bce3357u, 0x7c0f9e16u, 0x8ee7eb4au, 0xfe1a7f9bu, 0x4fe342e2u};   // curve_nist_p256.h:30-32
  static constexpr uint32_t MPRIME  = 0x00000001u;                                                                                                 // mgry_mul.h:37
};

template <> struct curve_consts<CURVE_SECP256K1> {
  // p = 2^256 - 2^32 - 977                                     tests/mgry.cpp:25-27
  static constexpr uint32_t P[8]    = {0xfffffc2fu, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  static constexpr uint32_t R_P[8]  = {0x000003d1u, 0x00000001u, 0, 0, 0, 0, 0, 0};
  static constexpr uint32_t RSQ[8]  = {0x000e90a1u, 0x000007a2u, 0x00000001u, 0, 0, 0, 0, 0};
  static constexpr uint32_t NEG_R[8] = {0xfffff85eu, 0xfffffffdu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  static constexpr uint32_t AM[8]   = {0, 0, 0, 0, 0, 0, 0, 0};
  static constexpr uint32_t BM[8]   = {0x00001ab7u, 0x00000007u, 0, 0, 0, 0, 0, 0};
  static constexpr uint32_t GX[8]   = {0x16f81798u, 0x59f2815bu, 0x2dce28d9u, 0x029bfcdbu, 0xce870b07u, 0x55a06295u, 0xf9dcbbacu, 0x79be667eu};
  static constexpr uint32_t GY[8]   = {0xfb10d4b8u, 0x9c47d08fu, 0xa6855419u, 0xfd17b448u, 0x0e1108a8u, 0x5da4fbfcu, 0x26a3c465u, 0x483ada77u};
  static constexpr uint32_t MPRIME  = 0xd2253531u;
};

template <> struct curve_consts<CURVE_SECP256K1_CLASSICAL> {
  static constexpr uint32_t P[8]    = {0xfffffc2fu, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  static constexpr uint32_t R_P[8]  = {1u, 0, 0, 0, 0, 0, 0, 0};      // "one"
  static constexpr uint32_t RSQ[8]  = {1u, 0, 0, 0, 0, 0, 0, 0};      // from_classical is the identity here
  static constexpr uint32_t NEG_R[8] = {0xfffffc2eu, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};   // -1
  static constexpr uint32_t AM[8]   = {0, 0, 0, 0, 0, 0, 0, 0};
  static constexpr uint32_t BM[8]   = {7u, 0, 0, 0, 0, 0, 0, 0};
  static constexpr uint32_t GX[8]   = {0x16f81798u, 0x59f2815bu, 0x2dce28d9u, 0x029bfcdbu, 0xce870b07u, 0x55a06295u, 0xf9dcbbacu, 0x79be667eu};
  static constexpr uint32_t GY[8]   = {0xfb10d4b8u, 0x9c47d08fu, 0xa6855419u, 0xfd17b448u, 0x0e1108a8u, 0x5da4fbfcu, 0x26a3c465u, 0x483ada77u};
  static constexpr uint32_t MPRIME  = 0u;                             // unused
};
template <> struct curve_consts<CURVE_P256_REFSQR> : curve_consts<CURVE_P256> {};
template <> struct curve_consts<CURVE_SECP256K1_REFSQR> : curve_consts<CURVE_SECP256K1> {};
template <int CURVE> struct curve_domain { static constexpr int fast = CURVE; };                       // domain the hot loops run in
template <> struct curve_domain<CURVE_SECP256K1> { static constexpr int fast = CURVE_SECP256K1_CLASSICAL; };
// which prime / which squaring an instance uses
template <int CURVE> struct curve_prime {
  static constexpr bool is_p256 = (CURVE == CURVE_P256 || CURVE == CURVE_P256_REFSQR);
  static constexpr bool ref_square = (CURVE == CURVE_P256_REFSQR || CURVE == CURVE_SECP256K1_REFSQR);
};

template <int CURVE, const uint32_t (&ARR)[8]> ECS_DEV fe fe_const() {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = ARR[i];
  return r;
}
#define FE_CONST(CURVE, NAME) fe_const<CURVE, curve_consts<CURVE>::NAME>()

// ---------------------------------------------------------------- global-memory access
// HBM layout: element i = 32 contiguous bytes (4 x u64 LE limbs).  Each lane moves its element
// with two 16-byte accesses; a wave covers 2 KiB contiguous, every fetched cache line is fully used.
ECS_DEV fe fe_load(const uint64_t* __restrict__ base, size_t i) {
  const uint4* p = reinterpret_cast<const uint4*>(base + 4 * i);
  uint4 lo = p[0], hi = p[1];
  fe r;
  r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w;
  r.w[4] = hi.x; r.w[5] = hi.y; r.w[6] = hi.z; r.w[7] = hi.w;
  return r;
}
ECS_DEV void fe_store(uint64_t* __restrict__ base, size_t i, const fe& v) {
  uint4* p = reinterpret_cast<uint4*>(base + 4 * i);
  p[0] = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
  p[1] = make_uint4(v.w[4], v.w[5], v.w[6], v.w[7]);
}
ECS_DEV fe2 fe2_load(const uint64_t* __restrict__ base, size_t i) {
  const uint4* p = reinterpret_cast<const uint4*>(base + 8 * i);
  fe2 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) { uint4 v = p[k]; r.w[4 * k] = v.x; r.w[4 * k + 1] = v.y; r.w[4 * k + 2] = v.z; r.w[4 * k + 3] = v.w; }
  return r;
}
ECS_DEV void fe2_store(uint64_t* __restrict__ base, size_t i, const fe2& v) {
  uint4* p = reinterpret_cast<uint4*>(base + 8 * i);
#pragma unroll
  for (int k = 0; k < 4; ++k) p[k] = make_uint4(v.w[4 * k], v.w[4 * k + 1], v.w[4 * k + 2], v.w[4 * k + 3]);
}

// ---------------------------------------------------------------- carry-chain primitives
// A lane mask in an SGPR pair: where the reference keeps carries / borrows as eve::logical masks
// (add.h:16-34), the natural gfx950 form is the carry-out of the last v_addc itself -- no VALU
// instruction is spent turning it into data.
typedef uint64_t lane_mask;

// r = a + b over 8 words; returns the carry-out as a lane mask.             add.h:11-34
// (three-operand form: when a stays live the compiler needs no register copies)
ECS_DEV lane_mask add8m3(fe& r, const fe& a, const fe& b) {
  lane_mask c;
  asm("v_add_co_u32 %0, vcc, %9, %17\n\t"
      "v_addc_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_addc_co_u32 %7, %8, %16, %24, vcc"
      : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]), "=&v"(r.w[6]), "=&v"(r.w[7]), "=&s"(c)
      : "v"(a.w[0]), "v"(a.w[1]), "v"(a.w[2]), "v"(a.w[3]), "v"(a.w[4]), "v"(a.w[5]), "v"(a.w[6]), "v"(a.w[7]),
        "v"(b.w[0]), "v"(b.w[1]), "v"(b.w[2]), "v"(b.w[3]), "v"(b.w[4]), "v"(b.w[5]), "v"(b.w[6]), "v"(b.w[7])
      : "vcc");
  return c;
}
// r = a - b, borrow returned as an all-ones / all-zeros word (three-operand form)
ECS_DEV uint32_t sub8_3(fe& r, const fe& a, const fe& b) {
  uint32_t m;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
      "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
      "v_subb_co_u32 %8, vcc, 0, 0, vcc"
      : "=&v"(r.w[0]), "=&v"(r.w[1]), "=&v"(r.w[2]), "=&v"(r.w[3]), "=&v"(r.w[4]), "=&v"(r.w[5]), "=&v"(r.w[6]), "=&v"(r.w[7]), "=&v"(m)
      : "v"(a.w[0]), "v"(a.w[1]), "v"(a.w[2]), "v"(a.w[3]), "v"(a.w[4]), "v"(a.w[5]), "v"(a.w[6]), "v"(a.w[7]),
        "v"(b.w[0]), "v"(b.w[1]), "v"(b.w[2]), "v"(b.w[3]), "v"(b.w[4]), "v"(b.w[5]), "v"(b.w[6]), "v"(b.w[7])
      : "vcc");
  return m;
}
// the same with the carry as a 0/1 word (callers that store it or add it)
ECS_DEV uint32_t add8(fe& a, const fe& b) {
  uint32_t c;
  asm("v_add_co_u32 %0, vcc, %0, %9\n\t"
      "v_addc_co_u32 %1, vcc, %1, %10, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, %11, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %12, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %4, %13, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %14, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %6, %15, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %7, %16, vcc\n\t"
      "v_addc_co_u32 %8, vcc, 0, 0, vcc"
      : "+v"(a.w[0]), "+v"(a.w[1]), "+v"(a.w[2]), "+v"(a.w[3]), "+v"(a.w[4]), "+v"(a.w[5]), "+v"(a.w[6]), "+v"(a.w[7]), "=v"(c)
      : "v"(b.w[0]), "v"(b.w[1]), "v"(b.w[2]), "v"(b.w[3]), "v"(b.w[4]), "v"(b.w[5]), "v"(b.w[6]), "v"(b.w[7])
      : "vcc");
  return c;
}
// a -= b over 8 words; returns the borrow as an all-ones / all-zeros word.  sub.h:12-38
ECS_DEV uint32_t sub8(fe& a, const fe& b) {
  uint32_t m;
  asm("v_sub_co_u32 %0, vcc, %0, %9\n\t"
      "v_subb_co_u32 %1, vcc, %1, %10, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %2, %11, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %3, %12, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %4, %13, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %5, %14, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %6, %15, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %7, %16, vcc\n\t"
      "v_subb_co_u32 %8, vcc, 0, 0, vcc"
      : "+v"(a.w[0]), "+v"(a.w[1]), "+v"(a.w[2]), "+v"(a.w[3]), "+v"(a.w[4]), "+v"(a.w[5]), "+v"(a.w[6]), "+v"(a.w[7]), "=v"(m)
      : "v"(b.w[0]), "v"(b.w[1]), "v"(b.w[2]), "v"(b.w[3]), "v"(b.w[4]), "v"(b.w[5]), "v"(b.w[6]), "v"(b.w[7])
      : "vcc");
  return m;
}

// r = (r + top*2^256 >= p) ? r - p : r, for r + top*2^256 < 2p.               sub.h:46-69
// `top` (the 257th bit) is a lane mask.  Three forms, chosen by ECS_COND_SUB (all return the same canonical residue):
//   0  r - p over 8 words, keep r where (borrow and not top) -- one scalar s_andn2_b64 -- and 8 v_cndmask by that
//      SGPR-pair mask (4.3-4.9 cycles each at 2-4 waves per SIMD);
//   1  the same subtraction, then the 8 results are MOVED over r under EXEC = the lanes that take them: v_mov_b32 issues
//      at 2.6-3.6 cycles where v_cndmask takes a full-rate slot (profiles/r02/valu_issue_rates_gfx950.txt);
//   2  the lanes with `top` set subtract p IN PLACE under EXEC (8 instructions, no temporaries, no selects); a lane
//      without it needs the subtraction only when r >= p, which needs r[7] == 0xffffffff (both primes' top word) --
//      2^-32 of the values -- so one v_cmp guards a wave-uniform branch to form 0.  Not constant-time in that 2^-32
//      case: kept for the public-input kernels, never the ladder's default (INTEGRATION.md "timing").
#ifndef ECS_COND_SUB
#define ECS_COND_SUB 1
#endif
template <int CURVE> ECS_DEV void cond_sub_p_select(fe& r, lane_mask top) {
  using K = curve_consts<CURVE>;
  fe d;
  lane_mask keep;
  if constexpr (curve_prime<CURVE>::is_p256) {
    // p's words are the inline constants -1, -1, -1, 0, 0, 0, 1, -1
    asm("v_sub_co_u32 %0, vcc, %9, -1\n\t"
        "v_subb_co_u32 %1, vcc, %10, -1, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %11, -1, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, 0, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %13, 0, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %14, 0, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, 1, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %16, -1, vcc\n\t"
        "s_andn2_b64 %8, vcc, %17"
        : "=&v"(d.w[0]), "=&v"(d.w[1]), "=&v"(d.w[2]), "=&v"(d.w[3]), "=&v"(d.w[4]), "=&v"(d.w[5]), "=&v"(d.w[6]), "=&v"(d.w[7]), "=&s"(keep)
        : "v"(r.w[0]), "v"(r.w[1]), "v"(r.w[2]), "v"(r.w[3]), "v"(r.w[4]), "v"(r.w[5]), "v"(r.w[6]), "v"(r.w[7]), "s"(top)
        : "vcc", "scc");
  } else {
    // secp256k1: words 2..7 are -1; words 0, 1 come in VGPRs (an SGPR source next to the VCC carry-in
    // would exceed gfx9's one-scalar-operand constant-bus limit)
    const uint32_t p0 = K::P[0], p1 = K::P[1];
    asm("v_sub_co_u32 %0, vcc, %9, %18\n\t"
        "v_subb_co_u32 %1, vcc, %10, %19, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %11, -1, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, -1, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %13, -1, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %14, -1, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, -1, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %16, -1, vcc\n\t"
        "s_andn2_b64 %8, vcc, %17"
        : "=&v"(d.w[0]), "=&v"(d.w[1]), "=&v"(d.w[2]), "=&v"(d.w[3]), "=&v"(d.w[4]), "=&v"(d.w[5]), "=&v"(d.w[6]), "=&v"(d.w[7]), "=&s"(keep)
        : "v"(r.w[0]), "v"(r.w[1]), "v"(r.w[2]), "v"(r.w[3]), "v"(r.w[4]), "v"(r.w[5]), "v"(r.w[6]), "v"(r.w[7]), "s"(top), "v"(p0), "v"(p1)
        : "vcc", "scc");
  }
  // keep set  <=>  value < p  <=>  keep r
#pragma unroll
  for (int i = 0; i < 8; ++i)
    asm("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r.w[i]) : "v"(d.w[i]), "s"(keep));
}
// form 1: d = r - p, then r <- d under EXEC = (top or no borrow)
template <int CURVE> ECS_DEV void cond_sub_p_move(fe& r, lane_mask top) {
  using K = curve_consts<CURVE>;
  fe d;
  lane_mask save;
#define ECS_MOVE_TAIL \
        "s_orn2_b64 vcc, %17, vcc\n\t"                 /* take d where top or not borrow */ \
        "s_and_saveexec_b64 %16, vcc\n\t" \
        "v_mov_b32 %8, %0\n\tv_mov_b32 %9, %1\n\tv_mov_b32 %10, %2\n\tv_mov_b32 %11, %3\n\t" \
        "v_mov_b32 %12, %4\n\tv_mov_b32 %13, %5\n\tv_mov_b32 %14, %6\n\tv_mov_b32 %15, %7\n\t" \
        "s_mov_b64 exec, %16"
#define ECS_MOVE_OUTS "=&v"(d.w[0]), "=&v"(d.w[1]), "=&v"(d.w[2]), "=&v"(d.w[3]), "=&v"(d.w[4]), "=&v"(d.w[5]), "=&v"(d.w[6]), "=&v"(d.w[7]), \
        "+v"(r.w[0]), "+v"(r.w[1]), "+v"(r.w[2]), "+v"(r.w[3]), "+v"(r.w[4]), "+v"(r.w[5]), "+v"(r.w[6]), "+v"(r.w[7]), "=&s"(save)
  if constexpr (curve_prime<CURVE>::is_p256) {
    asm("v_sub_co_u32 %0, vcc, %8, -1\n\t"
        "v_subb_co_u32 %1, vcc, %9, -1, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %10, -1, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %11, 0, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %12, 0, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %13, 0, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %14, 1, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %15, -1, vcc\n\t"
        ECS_MOVE_TAIL
        : ECS_MOVE_OUTS : "s"(top) : "vcc", "scc");
  } else {
    const uint32_t p0 = K::P[0], p1 = K::P[1];
    asm("v_sub_co_u32 %0, vcc, %8, %18\n\t"
        "v_subb_co_u32 %1, vcc, %9, %19, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %10, -1, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %11, -1, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %12, -1, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %13, -1, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %14, -1, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %15, -1, vcc\n\t"
        ECS_MOVE_TAIL
        : ECS_MOVE_OUTS : "s"(top), "v"(p0), "v"(p1) : "vcc", "scc");
  }
#undef ECS_MOVE_TAIL
#undef ECS_MOVE_OUTS
}
// form 2: in place under EXEC = top, then the 2^-32 guard
template <int CURVE> ECS_DEV void cond_sub_p_guard(fe& r, lane_mask top) {
  using K = curve_consts<CURVE>;
  lane_mask save, maybe;
  if constexpr (curve_prime<CURVE>::is_p256) {
    asm("s_and_saveexec_b64 %8, %10\n\t"
        "v_sub_co_u32 %0, vcc, %0, -1\n\t"
        "v_subb_co_u32 %1, vcc, %1, -1, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %2, -1, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %3, 0, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %4, 0, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %5, 0, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %6, 1, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %7, -1, vcc\n\t"
        "s_mov_b64 exec, %8\n\t"
        "v_cmp_eq_u32_e64 %9, %7, -1"
        : "+v"(r.w[0]), "+v"(r.w[1]), "+v"(r.w[2]), "+v"(r.w[3]), "+v"(r.w[4]), "+v"(r.w[5]), "+v"(r.w[6]), "+v"(r.w[7]), "=&s"(save), "=&s"(maybe)
        : "s"(top) : "vcc", "scc");
  } else {
    const uint32_t p0 = K::P[0], p1 = K::P[1];
    asm("s_and_saveexec_b64 %8, %10\n\t"
        "v_sub_co_u32 %0, vcc, %0, %11\n\t"
        "v_subb_co_u32 %1, vcc, %1, %12, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %2, -1, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %3, -1, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %4, -1, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %5, -1, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %6, -1, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %7, -1, vcc\n\t"
        "s_mov_b64 exec, %8\n\t"
        "v_cmp_eq_u32_e64 %9, %7, -1"
        : "+v"(r.w[0]), "+v"(r.w[1]), "+v"(r.w[2]), "+v"(r.w[3]), "+v"(r.w[4]), "+v"(r.w[5]), "+v"(r.w[6]), "+v"(r.w[7]), "=&s"(save), "=&s"(maybe)
        : "s"(top), "v"(p0), "v"(p1) : "vcc", "scc");
  }
  // A lane that subtracted is below p now (its r[7] may still read 0xffffffff: the full form leaves it alone).
  if (__builtin_expect(maybe != 0, 0)) cond_sub_p_select<CURVE>(r, 0);
}
template <int CURVE> ECS_DEV void cond_sub_p(fe& r, lane_mask top) {
#if ECS_COND_SUB == 2
  cond_sub_p_guard<CURVE>(r, top);
#elif ECS_COND_SUB == 1
  cond_sub_p_move<CURVE>(r, top);
#else
  cond_sub_p_select<CURVE>(r, top);
#endif
}
// lane mask of (v != 0)
ECS_DEV lane_mask mask_nonzero(uint32_t v) {
  lane_mask m;
  asm("v_cmp_ne_u32_e64 %0, %1, 0" : "=s"(m) : "v"(v));
  return m;
}

// ---------------------------------------------------------------- modular linear ops
// (a + b) mod p                                                            modular.h:10-15
template <int CURVE> ECS_DEV fe fe_add(const fe& x, const fe& b) {
  fe a;
  const lane_mask c = add8m3(a, x, b);
  cond_sub_p<CURVE>(a, c);
  return a;
}
// (a - b) mod p: subtract, then add p where the subtraction borrowed.        modular.h:24-41
// The add-back runs under EXEC = the borrow mask with p's words as inline constants: no mask word is ever built
// (16 VALU instructions + 2 scalar ones; the masked-operand form took 18).
#ifndef ECS_SUB_EXEC
#define ECS_SUB_EXEC 1
#endif
#if !ECS_SUB_EXEC
// the masked-operand form (round 1): borrow word, p & mask, one more chain.  Kept for the A/B: 47.8-47.9 M scalar mults/s against
// 48.5-48.6 with the EXEC form (-DECS_SUB_EXEC=0, same box, two runs each).
template <int CURVE> ECS_DEV fe fe_sub(const fe& x, const fe& b) {
  fe a;
  uint32_t m = sub8_3(a, x, b);
  if constexpr (curve_prime<CURVE>::is_p256) {
    uint32_t m1 = m & 1u;
    asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32 %1, vcc, %1, %8, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\t"
        "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %9, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %8, vcc"
        : "+v"(a.w[0]), "+v"(a.w[1]), "+v"(a.w[2]), "+v"(a.w[3]), "+v"(a.w[4]), "+v"(a.w[5]), "+v"(a.w[6]), "+v"(a.w[7])
        : "v"(m), "v"(m1) : "vcc");
  } else {
    using K = curve_consts<CURVE>;
    uint32_t m0 = m & K::P[0], m1 = m & K::P[1];
    asm("v_add_co_u32 %0, vcc, %0, %9\n\t"
        "v_addc_co_u32 %1, vcc, %1, %10, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %8, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %8, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %8, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %8, vcc"
        : "+v"(a.w[0]), "+v"(a.w[1]), "+v"(a.w[2]), "+v"(a.w[3]), "+v"(a.w[4]), "+v"(a.w[5]), "+v"(a.w[6]), "+v"(a.w[7])
        : "v"(m), "v"(m0), "v"(m1) : "vcc");
  }
  return a;
}
#else
template <int CURVE> ECS_DEV fe fe_sub(const fe& x, const fe& b) {
  fe a;
  lane_mask save;
#define ECS_SUB_HEAD \
      "v_sub_co_u32 %0, vcc, %9, %17\n\t" \
      "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t" \
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t" \
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t" \
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t" \
      "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t" \
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t" \
      "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t" \
      "s_and_saveexec_b64 %8, vcc\n\t"
#define ECS_SUB_OPS \
      : "=&v"(a.w[0]), "=&v"(a.w[1]), "=&v"(a.w[2]), "=&v"(a.w[3]), "=&v"(a.w[4]), "=&v"(a.w[5]), "=&v"(a.w[6]), "=&v"(a.w[7]), "=&s"(save) \
      : "v"(x.w[0]), "v"(x.w[1]), "v"(x.w[2]), "v"(x.w[3]), "v"(x.w[4]), "v"(x.w[5]), "v"(x.w[6]), "v"(x.w[7]), \
        "v"(b.w[0]), "v"(b.w[1]), "v"(b.w[2]), "v"(b.w[3]), "v"(b.w[4]), "v"(b.w[5]), "v"(b.w[6]), "v"(b.w[7])
  if constexpr (curve_prime<CURVE>::is_p256) {
    asm(ECS_SUB_HEAD
        "v_add_co_u32 %0, vcc, -1, %0\n\t"
        "v_addc_co_u32 %1, vcc, -1, %1, vcc\n\t"
        "v_addc_co_u32 %2, vcc, -1, %2, vcc\n\t"
        "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
        "v_addc_co_u32 %6, vcc, 1, %6, vcc\n\t"
        "v_addc_co_u32 %7, vcc, -1, %7, vcc\n\t"
        "s_mov_b64 exec, %8"
        ECS_SUB_OPS : "vcc", "scc");
  } else {
    using K = curve_consts<CURVE>;
    const uint32_t p0 = K::P[0], p1 = K::P[1];     // p = {p0, p1, -1, -1, -1, -1, -1, -1}
    asm(ECS_SUB_HEAD
        "v_add_co_u32 %0, vcc, %0, %25\n\t"
        "v_addc_co_u32 %1, vcc, %1, %26, vcc\n\t"
        "v_addc_co_u32 %2, vcc, -1, %2, vcc\n\t"
        "v_addc_co_u32 %3, vcc, -1, %3, vcc\n\t"
        "v_addc_co_u32 %4, vcc, -1, %4, vcc\n\t"
        "v_addc_co_u32 %5, vcc, -1, %5, vcc\n\t"
        "v_addc_co_u32 %6, vcc, -1, %6, vcc\n\t"
        "v_addc_co_u32 %7, vcc, -1, %7, vcc\n\t"
        "s_mov_b64 exec, %8"
        ECS_SUB_OPS, "v"(p0), "v"(p1) : "vcc", "scc");
  }
#undef ECS_SUB_HEAD
#undef ECS_SUB_OPS
  return a;
}
#endif
// 2a mod p                                                                 modular.h:17-22
template <int CURVE> ECS_DEV fe fe_dbl(const fe& a) {
  fe s;
  lane_mask c;                                                         // bit 255 of a = the carry out of the shift
  asm("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(c) : "v"(a.w[7]));
#pragma unroll
  for (int i = 7; i > 0; --i) s.w[i] = __builtin_amdgcn_alignbit(a.w[i], a.w[i - 1], 31);   // (a[i]:a[i-1]) >> 31
  s.w[0] = a.w[0] << 1;
  cond_sub_p<CURVE>(s, c);
  return s;
}
// 4a mod p in one pass (the reference doubles twice, mgry_ops.h:14-22: same canonical residue).  4a = q*2^256 + s with
// q = a >> 254 in 0..3 and s the low 256 bits; 2^256 = c (mod p), c = 2^256 - p, so 4a = s + q*c, which is below 2p:
// ONE conditional subtraction instead of two.
//   P-256:     c = 2^224 - 2^192 - 2^96 + 1.  s + q*c = s - X + q*2^224 with X = q*2^192 + q*2^96 - q, whose words are
//              {-q, m, m, q + m, 0, 0, q, 0} (m = all-ones iff q > 0); the + q*2^224 rides the same borrow chain as
//              "- (-q)" in word 7, and the 257th bit comes out as (q > 0) and not the final borrow.
//   secp256k1: c = 2^32 + 977: add {977 q, q} to the two low words; the 257th bit is the carry.
template <int CURVE> ECS_DEV fe fe_shl2(const fe& a) {
  fe s;
#pragma unroll
  for (int i = 7; i > 0; --i) s.w[i] = __builtin_amdgcn_alignbit(a.w[i], a.w[i - 1], 30);   // (a[i]:a[i-1]) >> 30
  s.w[0] = a.w[0] << 2;
  const uint32_t q = a.w[7] >> 30;
  lane_mask top;
  if constexpr (curve_prime<CURVE>::is_p256) {
    uint32_t nq, m, x3;
    asm("v_sub_co_u32 %8, %11, 0, %12\n\t"            /* nq = -q, %11 = (q > 0) */
        "v_subb_co_u32 %9, vcc, 0, 0, %11\n\t"          /* m = -(q > 0) */
        "v_add_u32 %10, %12, %9\n\t"                    /* x3 = q + m */
        "v_sub_co_u32 %0, vcc, %0, %8\n\t"
        "v_subb_co_u32 %1, vcc, %1, %9, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %2, %9, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %3, %10, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %4, 0, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %5, 0, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %6, %12, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %7, %8, vcc\n\t"        /* - (-q) = + q at 2^224 */
        "s_andn2_b64 %11, %11, vcc"                     /* 257th bit = (q > 0) and not borrow */
        : "+v"(s.w[0]), "+v"(s.w[1]), "+v"(s.w[2]), "+v"(s.w[3]), "+v"(s.w[4]), "+v"(s.w[5]), "+v"(s.w[6]), "+v"(s.w[7]),
          "=&v"(nq), "=&v"(m), "=&v"(x3), "=&s"(top)
        : "v"(q) : "vcc", "scc");
  } else {
    const uint32_t k977 = 977u;
    uint32_t lo;
    asm("v_mul_u32_u24 %8, %10, %11\n\t"               /* 977 q */
        "v_add_co_u32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32 %1, vcc, %1, %10, vcc\n\t"
        "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
        "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
        "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
        "v_addc_co_u32 %7, %9, 0, %7, vcc"
        : "+v"(s.w[0]), "+v"(s.w[1]), "+v"(s.w[2]), "+v"(s.w[3]), "+v"(s.w[4]), "+v"(s.w[5]), "+v"(s.w[6]), "+v"(s.w[7]), "=&v"(lo), "=&s"(top)
        : "v"(q), "v"(k977) : "vcc");
  }
  cond_sub_p<CURVE>(s, top);
  return s;
}
// a * 2^N mod p = N successive doublings (pairs of them in one pass)         mgry_ops.h:14-22
template <int CURVE, int N> ECS_DEV fe fe_shl(fe a) {
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2) a = fe_shl2<CURVE>(a);
  if constexpr (N & 1) a = fe_dbl<CURVE>(a);
  return a;
}
// gfp.h:60-64 opposite(): (-R) - (a - R), two modular subtractions.  Equal to fe_neg for canonical a; kept in
// the reference's two-step form so that even an unreduced operand (a >= p) yields the reference's bits.
template <int CURVE> ECS_DEV fe fe_opposite(const fe& a) {
  return fe_sub<CURVE>(FE_CONST(CURVE, NEG_R), fe_sub<CURVE>(a, FE_CONST(CURVE, R_P)));
}
// -a mod p for canonical a (0 stays 0).
template <int CURVE> ECS_DEV fe fe_neg(const fe& a) {
  fe z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z.w[i] = 0;
  return fe_sub<CURVE>(z, a);
}

// ---------------------------------------------------------------- 256 x 256 -> 512 multiply
// Product scanning (Comba) on 32-bit words with a 96-bit column accumulator: each partial
// product is one v_mad_u64_u32 (64-bit accumulate, carry-out in VCC) + one v_addc_co_u32 into
// the third accumulator word.  Replaces mul.h:115-158 (64 vpmuludq + digit renormalisation).
ECS_DEV void mac_nocarry(uint64_t& acc, uint32_t a, uint32_t b) {                // sum provably < 2^64
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
ECS_DEV uint64_t mul_wide(uint32_t a, uint32_t b) {
  uint64_t r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b) : "vcc");
  return r;
}

// One Comba column in ONE asm statement: acc += sum_i a[i]*b[i], ex = the number of carries out of the
// 64-bit accumulator.  (The compiler separates dependent asm statements by an s_nop on gfx950; a statement per
// partial product paid 64 of them per multiply.)
#define ECS_MAC0(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc\n\t"
#define ECS_MAC(A, B)  "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
template <int N> ECS_DEV void mac_col(uint64_t& acc, uint32_t& ex, const uint32_t (&a)[N], const uint32_t (&b)[N]) {
  static_assert(N >= 1 && N <= 8, "column length");
  if constexpr (N == 1)
    asm(ECS_MAC0(2, 3) : "+v"(acc), "=&v"(ex) : "v"(a[0]), "v"(b[0]) : "vcc");
  else if constexpr (N == 2)
    asm(ECS_MAC0(2, 4) ECS_MAC(3, 5) : "+v"(acc), "=&v"(ex) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]) : "vcc");
  else if constexpr (N == 3)
    asm(ECS_MAC0(2, 5) ECS_MAC(3, 6) ECS_MAC(4, 7) : "+v"(acc), "=&v"(ex)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]) : "vcc");
  else if constexpr (N == 4)
    asm(ECS_MAC0(2, 6) ECS_MAC(3, 7) ECS_MAC(4, 8) ECS_MAC(5, 9) : "+v"(acc), "=&v"(ex)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "vcc");
  else if constexpr (N == 5)
    asm(ECS_MAC0(2, 7) ECS_MAC(3, 8) ECS_MAC(4, 9) ECS_MAC(5, 10) ECS_MAC(6, 11) : "+v"(acc), "=&v"(ex)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]) : "vcc");
  else if constexpr (N == 6)
    asm(ECS_MAC0(2, 8) ECS_MAC(3, 9) ECS_MAC(4, 10) ECS_MAC(5, 11) ECS_MAC(6, 12) ECS_MAC(7, 13) : "+v"(acc), "=&v"(ex)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]) : "vcc");
  else if constexpr (N == 7)
    asm(ECS_MAC0(2, 9) ECS_MAC(3, 10) ECS_MAC(4, 11) ECS_MAC(5, 12) ECS_MAC(6, 13) ECS_MAC(7, 14) ECS_MAC(8, 15) : "+v"(acc), "=&v"(ex)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]) : "vcc");
  else
    asm(ECS_MAC0(2, 10) ECS_MAC(3, 11) ECS_MAC(4, 12) ECS_MAC(5, 13) ECS_MAC(6, 14) ECS_MAC(7, 15) ECS_MAC(8, 16) ECS_MAC(9, 17) : "+v"(acc), "=&v"(ex)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
}
// column k of a*b: the products a[i]*b[k-i], lo <= i <= hi
template <int K> ECS_DEV void mul_col(uint64_t& acc, uint32_t& ex, const fe& a, const fe& b) {
  constexpr int lo = K > 7 ? K - 7 : 0, hi = K < 7 ? K : 7, N = hi - lo + 1;
  uint32_t x[N], y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { x[i] = a.w[lo + i]; y[i] = b.w[K - lo - i]; }
  mac_col<N>(acc, ex, x, y);
}

ECS_DEV fe2 mul8x8(const fe& a, const fe& b) {
  fe2 t;
  uint64_t acc = mul_wide(a.w[0], b.w[0]);
  t.w[0] = (uint32_t)acc;
  acc >>= 32;
  uint32_t ex;
#define ECS_COL(K) mul_col<K>(acc, ex, a, b); t.w[K] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)ex << 32);
  // column 1: a0*b1 lands on hi(a0*b0) < 2^32 and cannot carry; only a1*b0 can
  asm("v_mad_u64_u32 %0, vcc, %2, %5, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, 0, vcc"
      : "+v"(acc), "=&v"(ex) : "v"(a.w[0]), "v"(a.w[1]), "v"(b.w[0]), "v"(b.w[1]) : "vcc");
  t.w[1] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)ex << 32);
  ECS_COL(2) ECS_COL(3) ECS_COL(4) ECS_COL(5) ECS_COL(6) ECS_COL(7)
  ECS_COL(8) ECS_COL(9) ECS_COL(10) ECS_COL(11) ECS_COL(12) ECS_COL(13)
#undef ECS_COL
  mac_nocarry(acc, a.w[7], b.w[7]);                                 // top column: total < 2^512
  t.w[14] = (uint32_t)acc;
  t.w[15] = (uint32_t)(acc >> 32);
  return t;
}

// column k of the cross products of a^2: a[i]*a[k-i] for i < k-i <= 7
template <int K> ECS_DEV void sqr_col(uint64_t& acc, uint32_t& ex, const fe& a) {
  constexpr int lo = K > 7 ? K - 7 : 0, hi = (K - 1) / 2, N = hi - lo + 1;
  uint32_t x[N], y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { x[i] = a.w[lo + i]; y[i] = a.w[K - lo - i]; }
  mac_col<N>(acc, ex, x, y);
}

// Squaring: 28 cross products, doubled as a whole, plus the 8 diagonal squares.  mul.h:160-221
ECS_DEV fe2 sqr8(const fe& a) {
  fe2 t;
  // cross = sum_{i<j} a_i a_j 2^(32(i+j)), columns 1..13
  uint32_t c[16];
  c[0] = 0;
  uint64_t acc = mul_wide(a.w[0], a.w[1]);
  uint32_t ex;
  c[1] = (uint32_t)acc; acc >>= 32;
  mac_nocarry(acc, a.w[0], a.w[2]);                                  // a*b + (< 2^32) cannot overflow
  c[2] = (uint32_t)acc; acc >>= 32;
#define ECS_SQCOL(K) sqr_col<K>(acc, ex, a); c[K] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)ex << 32);
  ECS_SQCOL(3) ECS_SQCOL(4) ECS_SQCOL(5) ECS_SQCOL(6) ECS_SQCOL(7) ECS_SQCOL(8)
  ECS_SQCOL(9) ECS_SQCOL(10) ECS_SQCOL(11) ECS_SQCOL(12)
#undef ECS_SQCOL
  mac_nocarry(acc, a.w[6], a.w[7]);  // cross < 2^479: columns 13 and 14 together are < 2^63, the last product cannot carry
  c[13] = (uint32_t)acc;
  c[14] = (uint32_t)(acc >> 32);    // word 14 is the top word
  // double: (c << 1), 16 words
  c[15] = c[14] >> 31;
#pragma unroll
  for (int i = 14; i > 0; --i) c[i] = __builtin_amdgcn_alignbit(c[i], c[i - 1], 31);
  // c[0] stays 0.  Add the diagonal squares d_i = a_i^2 at word 2i: two 32-bit carry chains
  // (words 1..7, then 8..15); the carry between them travels as an SGPR lane mask.
  uint64_t d[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = mul_wide(a.w[i], a.w[i]);
  t.w[0] = (uint32_t)d[0];
  uint32_t dh[8], dl[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dl[i] = (uint32_t)d[i]; dh[i] = (uint32_t)(d[i] >> 32); }
  lane_mask cy;                             // carry from word 7 into word 8, kept as a lane mask
  asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
      "v_addc_co_u32 %6, %7, %6, %14, vcc"
      : "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "=&s"(cy)
      : "v"(dh[0]), "v"(dl[1]), "v"(dh[1]), "v"(dl[2]), "v"(dh[2]), "v"(dl[3]), "v"(dh[3])
      : "vcc");
#pragma unroll
  for (int i = 4; i < 8; ++i) d[i] = mul_wide(a.w[i], a.w[i]);
#pragma unroll
  for (int i = 4; i < 8; ++i) { dl[i] = (uint32_t)d[i]; dh[i] = (uint32_t)(d[i] >> 32); }
  asm("v_addc_co_u32 %0, vcc, %0, %8, %16\n\t"        // carry-in: the lane mask of the first chain
      "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %7, %15, vcc"
      : "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15])
      : "v"(dl[4]), "v"(dh[4]), "v"(dl[5]), "v"(dh[5]), "v"(dl[6]), "v"(dh[6]), "v"(dl[7]), "v"(dh[7]), "s"(cy)
      : "vcc");
#pragma unroll
  for (int i = 1; i < 16; ++i) t.w[i] = c[i];
  return t;
}

// The reference's square() AS WRITTEN (mul.h:160-212 square_u32_zext between zext_u32x64 and trunc_u64x32): operand
// scanning on 32-bit digits held in 64-bit lanes, the cross products doubled one by one.  At mul.h:186-190 the sum
// 2*a_i*a_j + ret + prevs[0] can exceed 2^64 and the lane wraps (the reference's "TODO: carry?", mul.h:207); the
// digits are truncated to 32 bits at the end (mul.h:100-107).  This returns the reference's bits on every input --
// including the ones where they are not a^2.  Only the ECSIMD_HIP_REF_SQUARE_COMPAT instances use it
// (curve_prime<C>::ref_square).
//
// sqr8_ref_c is the restatement in wrapping uint64_t C (round 2; proven equal to the compiled reference on 200 000
// carry-heavy operands): ~230 instructions as the compiler lowers it.  sqr8_ref is the same function of a, hand-laid for
// gfx950 (round 3, ~150 instructions), one "node" (i, j) = one doubled cross product:
//   * c = bit 63 of a_i*a_j (mul.h:184 `carry`) is the CARRY-OUT of the multiply itself when 2^63 is the accumulator
//     input -- v_mad_u64_u32 writes it to an SGPR pair as a lane mask -- and the result's flipped bit 63 is the bit the
//     doubling shifts out anyway;
//   * u = 2*P + (ret + prevs0) mod 2^64 (mul.h:185-187) is ONE v_lshl_add_u64: a 64-bit add with no carry-out, which is
//     exactly the reference's dropped carry;
//   * the next node's addend ret' + prevs1 + (u >> 32) (mul.h:189-192: the digit below is renormalised, prevs shift) is a
//     two-word carry chain whose carry-IN is the lane mask of the node before.
// The digit a row leaves unnormalised (mul.h:207 `ret[i+nlimbs] += prevs[0]`) stays a 64-bit value until the next row's
// last node (or the last diagonal) consumes it, as in the reference.
ECS_DEV fe2 sqr8_ref_c(const fe& a) {
  uint64_t ret[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ret[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t t = (uint64_t)a.w[i] * a.w[i];
    t += ret[2 * i];
    ret[2 * i] = t & 0xffffffffu;
    uint64_t prevs0 = t >> 32, prevs1 = 0;
#pragma unroll
    for (int j = i + 1; j < 8; ++j) {
      uint64_t u = (uint64_t)a.w[i] * a.w[j];
      const uint64_t carry = u >> 63;
      u <<= 1;
      u += ret[i + j];
      u += prevs0;                         // may wrap: the reference's dropped carry
      ret[i + j] = u & 0xffffffffu;
      prevs0 = prevs1;
      prevs0 += u >> 32;
      prevs1 = carry;
    }
    ret[i + 8] += prevs0;                  // mul.h:207
    if (i + 9 < 16) ret[i + 9] = prevs1;
  }
  fe2 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r.w[i] = (uint32_t)ret[i];
  return r;
}

// One node: u = (2*a*b + S) mod 2^64 and c = the lanes where a*b >= 2^63.  The multiply accumulates onto 2^63: its carry-out
// IS c, and the bit it flips is the one the doubling shifts out.  (One statement: a compiler-scheduled instruction that
// reads the result of an asm statement is held back by an s_nop -- the gfx950 trans-forwarding hazard it cannot rule out.)
ECS_DEV uint64_t node_ref(uint32_t a, uint32_t b, uint64_t S, lane_mask& c) {
  uint64_t u;
  const uint64_t k63 = 0x8000000000000000ull;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %5\n\t"
      "v_lshl_add_u64 %0, %0, 1, %4"
      : "=&v"(u), "=&s"(c) : "v"(a), "v"(b), "v"(S), "s"(k63));
  return u;
}
// x + y + (hi << 32) [+ 1 where cin], hi in {absent = 0}: the next node's 64-bit addend from 32-bit pieces
ECS_DEV uint64_t sum33(uint32_t x, uint32_t y) {
  uint32_t lo, hi;
  asm("v_add_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc" : "=&v"(lo), "=&v"(hi) : "v"(x), "v"(y) : "vcc");
  return ((uint64_t)hi << 32) | lo;
}
ECS_DEV uint64_t sum33c(uint32_t x, uint32_t y, lane_mask cin) {
  uint32_t lo, hi;
  asm("v_addc_co_u32 %0, vcc, %2, %3, %4\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc" : "=&v"(lo), "=&v"(hi) : "v"(x), "v"(y), "s"(cin) : "vcc");
  return ((uint64_t)hi << 32) | lo;
}
// the same with a 64-bit x (the unnormalised digit)
ECS_DEV uint64_t sum33w(uint64_t x, uint32_t y) {
  uint32_t lo, hi;
  asm("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, 0, %3, vcc" : "=&v"(lo), "=&v"(hi) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "v"(y) : "vcc");
  return ((uint64_t)hi << 32) | lo;
}
ECS_DEV uint64_t sum33wc(uint64_t x, uint32_t y, lane_mask cin) {
  uint32_t lo, hi;
  asm("v_addc_co_u32 %0, vcc, %2, %4, %5\n\tv_addc_co_u32 %1, vcc, 0, %3, vcc" : "=&v"(lo), "=&v"(hi) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "v"(y), "s"(cin) : "vcc");
  return ((uint64_t)hi << 32) | lo;
}
// x + [c1] + [c2] as a 64-bit value (x a 32-bit word, c1 and c2 lane masks)
ECS_DEV uint64_t sum_two_carries(uint32_t x, lane_mask c1, lane_mask c2) {
  uint32_t lo, hi;
  lane_mask k;
  asm("v_addc_co_u32 %0, %2, %3, 0, %4\n\t"          /* x + c1, carry k */
      "v_addc_co_u32 %0, vcc, %0, 0, %5\n\t"          /* + c2, carry vcc: at most one of the two can carry */
      "s_or_b64 vcc, vcc, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, 0, vcc"
      : "=&v"(lo), "=&v"(hi), "=&s"(k) : "v"(x), "s"(c1), "s"(c2) : "vcc", "scc");
  return ((uint64_t)hi << 32) | lo;
}

ECS_DEV fe2 sqr8_ref(const fe& a) {
  uint32_t R[16];                       // normalised digits (mul.h:188 `& low_mask`)
  uint64_t W = 0;                       // the unnormalised digit at position i + 8 after row i (mul.h:207)
  lane_mask c9 = 0;                     // ret[i + 9] = prevs[1] after row i (mul.h:208-210): a 0/1 digit, kept as a lane mask
#pragma unroll
  for (int i = 0; i < 16; ++i) R[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // ---- the diagonal square joins digit 2i (mul.h:172-175); prevs[0] = its high half
    uint64_t S;                          // addend of the next node: ret[i + j] + prevs[0]
    if (i == 0) {
      const uint64_t t = mul_wide(a.w[0], a.w[0]);
      R[0] = (uint32_t)t;
      S = t >> 32;                       // ret[1] is still zero
    } else if (i < 7) {
      const uint64_t t = mul_wide(a.w[i], a.w[i]);      // < 2^64 - 2^33 + 2: adding a 32-bit digit cannot wrap
      uint32_t lo; lane_mask k;
      asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(lo), "=s"(k) : "v"((uint32_t)t), "v"(R[2 * i]));
      R[2 * i] = lo;
      // prevs[0] = (t + ret[2i]) >> 32 = t.hi + k; the next addend is ret[2i + 1] + prevs[0]
      if (i == 6) S = sum33wc(W, (uint32_t)(t >> 32), k);          // 2i + 1 = 13 = the unnormalised digit of row 5
      else S = sum33c(R[2 * i + 1], (uint32_t)(t >> 32), k);
    } else {
      // i = 7: digit 14 is row 6's unnormalised one; the 64-bit lane wraps like the reference's (mul.h:173-174)
      uint64_t t;
      asm("v_mad_u64_u32 %0, vcc, %1, %1, %2" : "=v"(t) : "v"(a.w[7]), "v"(W) : "vcc");
      R[14] = (uint32_t)t;
      uint32_t top;                      // ret[15] = prevs[1] of row 6 + (t >> 32), truncated to 32 bits (mul.h:100-107)
      asm("v_addc_co_u32 %0, vcc, %1, 0, %2" : "=v"(top) : "v"((uint32_t)(t >> 32)), "s"(c9) : "vcc");
      R[15] = top;
      break;
    }
    // ---- the doubled cross products of row i (mul.h:180-196)
    lane_mask cm1 = 0, cm2 = 0;          // carry of node j - 1 and of node j - 2 (prevs[1] and what moved into prevs[0])
    uint32_t uhi = 0;
#pragma unroll
    for (int j = i + 1; j < 8; ++j) {
      lane_mask c;
      const uint64_t u = node_ref(a.w[i], a.w[j], S, c);   // the 64-bit add wraps where the reference's lane wraps
      R[i + j] = (uint32_t)u;
      uhi = (uint32_t)(u >> 32);
      cm2 = cm1; cm1 = c;
      if (j < 7) {
        // addend of node j + 1: ret[i + j + 1] + (prevs[1] of node j - 1) + (u >> 32)
        const bool wide = (j + 1 == 7) && (i >= 1);               // position i + 7 holds row i - 1's unnormalised digit
        if (j == i + 1) S = wide ? sum33w(W, uhi) : sum33(R[i + j + 1], uhi);      // prevs[1] is still zero
        else S = wide ? sum33wc(W, uhi, cm2) : sum33c(R[i + j + 1], uhi, cm2);
      }
    }
    // ---- row end (mul.h:207-210): ret[i + 8] += prevs[0] = (carry of node 6) + (u_7 >> 32), on top of row i - 1's 0/1 digit;
    // ret[i + 9] = the carry of node 7
    if (i == 6) { W = (i >= 1) ? sum_two_carries(uhi, 0, c9) : 0; }   // row 6 has one node: its prevs[1] before it is zero
    else if (i == 0) { uint32_t lo, hi; asm("v_addc_co_u32 %0, vcc, %2, 0, %3\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc" : "=&v"(lo), "=&v"(hi) : "v"(uhi), "s"(cm2) : "vcc"); W = ((uint64_t)hi << 32) | lo; }
    else W = sum_two_carries(uhi, cm2, c9);
    c9 = cm1;
  }
  fe2 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r.w[i] = R[i];
  return r;
}

// ---------------------------------------------------------------- Montgomery reduction
// T * 2^-256 mod p, canonical.  Same value as mgry_mul.h:84-121 (8 rounds of 32-bit digits + one
// conditional subtract); the word size and the use of the primes' special form differ, the
// residue does not (Montgomery reduction is a function of T and p only).

// P-256: p = -1 mod 2^96, so m' = 1 (mgry_mul.h:37 gives mprime = 1) and q = the low 64 bits
// themselves.  q*p = q*2^256 - q*2^224 + q*2^192 + q*2^96 - q: no multiplies.  Four 64-bit
// rounds; in each, with M = q*(2^64 - 2^32 + 1) (the p[3] limb 0xffffffff00000001 times q):
//   t[o+3..o+4] += q,  t[o+6..o+9] += M,  carry out -> word o+10 (an SGPR lane mask, the carry-in of the next round's M).
ECS_DEV fe mgry_reduce_p256(fe2& t) {
  // The four rounds are ONE asm statement: the compiler puts an s_nop between an inline-asm statement and a following
  // instruction that reads a register it wrote (gfx950 forwarding-hazard workaround it cannot rule out for asm), and the
  // statement-per-chain form paid 12 of them per reduction.  %0..%12 = t[3..15], %13..%15 = M1..M3, %16 = the chain
  // carry (an SGPR pair: each round's carry-out is the next round's carry-in, the last one is the 257th bit),
  // %17..%19 = t[0..2].
  lane_mask cin;
  uint32_t m1, m2, m3;
  asm(
      "v_sub_co_u32 %13, vcc, %18, %17\n\t"
      "v_subb_co_u32 %14, vcc, %17, %18, vcc\n\t"
      "v_subb_co_u32 %15, vcc, %18, 0, vcc\n\t"
      "v_add_co_u32 %0, vcc, %0, %17\n\t"
      "v_addc_co_u32 %1, vcc, %1, %18, vcc\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %17, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %4, %13, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %14, vcc\n\t"
      "v_addc_co_u32 %6, %16, %6, %15, vcc\n\t"
      "v_sub_co_u32 %13, vcc, %0, %19\n\t"
      "v_subb_co_u32 %14, vcc, %19, %0, vcc\n\t"
      "v_subb_co_u32 %15, vcc, %0, 0, vcc\n\t"
      "v_addc_co_u32 %14, vcc, 0, %14, %16\n\t"
      "v_addc_co_u32 %15, vcc, 0, %15, vcc\n\t"
      "v_add_co_u32 %2, vcc, %2, %19\n\t"
      "v_addc_co_u32 %3, vcc, %3, %0, vcc\n\t"
      "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %19, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %6, %13, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %7, %14, vcc\n\t"
      "v_addc_co_u32 %8, %16, %8, %15, vcc\n\t"
      "v_sub_co_u32 %13, vcc, %2, %1\n\t"
      "v_subb_co_u32 %14, vcc, %1, %2, vcc\n\t"
      "v_subb_co_u32 %15, vcc, %2, 0, vcc\n\t"
      "v_addc_co_u32 %14, vcc, 0, %14, %16\n\t"
      "v_addc_co_u32 %15, vcc, 0, %15, vcc\n\t"
      "v_add_co_u32 %4, vcc, %4, %1\n\t"
      "v_addc_co_u32 %5, vcc, %5, %2, vcc\n\t"
      "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %7, %1, vcc\n\t"
      "v_addc_co_u32 %8, vcc, %8, %13, vcc\n\t"
      "v_addc_co_u32 %9, vcc, %9, %14, vcc\n\t"
      "v_addc_co_u32 %10, %16, %10, %15, vcc\n\t"
      "v_sub_co_u32 %13, vcc, %4, %3\n\t"
      "v_subb_co_u32 %14, vcc, %3, %4, vcc\n\t"
      "v_subb_co_u32 %15, vcc, %4, 0, vcc\n\t"
      "v_addc_co_u32 %14, vcc, 0, %14, %16\n\t"
      "v_addc_co_u32 %15, vcc, 0, %15, vcc\n\t"
      "v_add_co_u32 %6, vcc, %6, %3\n\t"
      "v_addc_co_u32 %7, vcc, %7, %4, vcc\n\t"
      "v_addc_co_u32 %8, vcc, 0, %8, vcc\n\t"
      "v_addc_co_u32 %9, vcc, %9, %3, vcc\n\t"
      "v_addc_co_u32 %10, vcc, %10, %13, vcc\n\t"
      "v_addc_co_u32 %11, vcc, %11, %14, vcc\n\t"
      "v_addc_co_u32 %12, %16, %12, %15, vcc"
      : "+v"(t.w[3]), "+v"(t.w[4]), "+v"(t.w[5]), "+v"(t.w[6]), "+v"(t.w[7]), "+v"(t.w[8]), "+v"(t.w[9]), "+v"(t.w[10]),
        "+v"(t.w[11]), "+v"(t.w[12]), "+v"(t.w[13]), "+v"(t.w[14]), "+v"(t.w[15]), "=&v"(m1), "=&v"(m2), "=&v"(m3), "=&s"(cin)
      : "v"(t.w[0]), "v"(t.w[1]), "v"(t.w[2])
      : "vcc");
  fe res;
#pragma unroll
  for (int i = 0; i < 8; ++i) res.w[i] = t.w[8 + i];
  cond_sub_p<CURVE_P256>(res, cin);
  return res;
}

// Generic word-serial reduction (any odd p given as constants): 8 rounds of
//   q = t[i] * m' mod 2^32;  t += q * p * 2^(32 i)
// with the row q*p accumulated by v_mad_u64_u32 (a*b + t + carry < 2^64, so no carry-out).
// Used for secp256k1 (m' = 0xd2253531).                                  mgry_mul.h:110-116
template <int CURVE> ECS_DEV fe mgry_reduce_generic(fe2& t) {
  using K = curve_consts<CURVE>;
  uint32_t top = 0;     // carries beyond the current row's last word
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t q = t.w[i] * K::MPRIME;
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // acc = q*p[j] + t[i+j] + carry  (fits 64 bits)
      uint64_t acc = (uint64_t)t.w[i + j] + carry;
      mac_nocarry(acc, q, K::P[j]);
      t.w[i + j] = (uint32_t)acc;
      carry = (uint32_t)(acc >> 32);
    }
    // add the row's carry (+ the previous round's overflow) into word i+8; what overflows belongs
    // to word i+9 and is consumed by the next round (after the last round it is the 257th bit)
    uint64_t s = (uint64_t)t.w[i + 8] + carry + top;
    t.w[i + 8] = (uint32_t)s;
    top = (uint32_t)(s >> 32);
  }
  fe res;
#pragma unroll
  for (int i = 0; i < 8; ++i) res.w[i] = t.w[8 + i];
  cond_sub_p<CURVE>(res, mask_nonzero(top));
  return res;
}

#ifndef ECS_K1_REDUCE_MAD64
#define ECS_K1_REDUCE_MAD64 1          // 0: round 1-4's borrow-tracking rounds (the A/B build)
#endif
// secp256k1, Montgomery domain: T * 2^-256 mod p with the prime's special form.  The reference's eight 32-bit rounds
// (mgry_mul.h:110-116) add q_i * p * W^i, W = 2^32; with p = W^8 - c, c = W + 977, that is  + q_i W^(8+i)  - q_i c W^i:
//     (T + Q p) / W^8  =  T_hi + Q - E,      Q = sum q_i W^i,   E = (Q c - T_lo) / W^8   (exact: Q c = T_lo mod W^8).
// Only the low half has to be walked word by word, and each round multiplies by the 10-bit constant 977 instead of by
// the 8 words of p: with D_i the amount still to be subtracted at word i (D_0 = 0, D_i < 2^34),
//     s = t_i - lo(D_i)  (borrow b);   q_i = s * m' mod W   [m' = 977^-1 mod W, so lo(977 q_i) = s: word i cancels];
//     D_(i+1) = hi(D_i) + b + hi(977 q_i) + q_i;                                    and E = D_8.
// 6 instructions per round (2 of them multiplies) + 16 for T_hi + Q - E: 64, against 224 for the generic row-by-row
// form -- same residue, Montgomery reduction being a function of T and p only.
ECS_DEV fe mgry_reduce_secp256k1(fe2& t) {
  const uint32_t MP = curve_consts<CURVE_SECP256K1>::MPRIME, K = 977u;
  fe q; uint32_t dl, dh;
#if ECS_K1_REDUCE_MAD64
  // Round 5 (the reference-square ladder of secp256k1 ran 8 % behind P-256's, and this reduction -- which those instances cannot trade
  // for the classical domain's fold, the dropped carry being a function of the MONTGOMERY digits -- is where: 64 instructions against P-256's 46, sixteen
  // times per iteration).  The same recurrence with the 64-bit D kept whole:  A = D_i + 977 q_i  is ONE v_mad_u64_u32; its low word IS t_i (that is what
  // q_i = (t_i - lo(D_i)) * m' means), so (A - t_i) / W = hi(A) with no borrow to track, and  D_(i+1) = hi(A) + q_i.  Five instructions per round, one of
  // them a full-rate v_sub_u32, where the borrow-tracking form has six half-rate ones.  Same q_i, same D_8: the same residue.
  uint64_t D = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t sw = t.w[i] - (uint32_t)D;
    q.w[i] = sw * MP;
    uint64_t A;                                             // (asm: left to itself the compiler adds hi(A) + q_i as two zero-extended 64-bit values, three moves a round)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(A) : "v"(q.w[i]), "s"(K), "v"(D) : "vcc");
    uint32_t lo, hi;
    asm("v_add_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc" : "=&v"(lo), "=v"(hi) : "v"((uint32_t)(A >> 32)), "v"(q.w[i]) : "vcc");
    D = ((uint64_t)hi << 32) | lo;
  }
  dl = (uint32_t)D; dh = (uint32_t)(D >> 32);
#else
  uint32_t sw, xh;
  // %0..%7 = q_0..q_7, %8 = lo(D), %9 = hi(D), %10 = s, %11 = hi(977 q), %12..%19 = t_0..t_7, %20 = m', %21 = 977
#define ECS_K1_ROUND(QI, TI) \
      "v_sub_co_u32 %10, vcc, " TI ", %8\n\t"        /* s = t_i - lo(D), borrow in VCC */ \
      "v_mul_lo_u32 " QI ", %10, %20\n\t"             /* q_i = s * m' */ \
      "v_mul_hi_u32 %11, " QI ", %21\n\t"             /* hi(977 q_i) */ \
      "v_addc_co_u32 %11, vcc, %9, %11, vcc\n\t"      /* + hi(D) + b   (< 2^11: no carry) */ \
      "v_add_co_u32 %8, vcc, %11, " QI "\n\t"         /* lo(D') = ... + q_i */ \
      "v_addc_co_u32 %9, vcc, 0, 0, vcc\n\t"          /* hi(D') */
  asm("v_mov_b32 %8, 0\n\tv_mov_b32 %9, 0\n\t"
      ECS_K1_ROUND("%0", "%12") ECS_K1_ROUND("%1", "%13") ECS_K1_ROUND("%2", "%14") ECS_K1_ROUND("%3", "%15")
      ECS_K1_ROUND("%4", "%16") ECS_K1_ROUND("%5", "%17") ECS_K1_ROUND("%6", "%18") ECS_K1_ROUND("%7", "%19")
      : "=&v"(q.w[0]), "=&v"(q.w[1]), "=&v"(q.w[2]), "=&v"(q.w[3]), "=&v"(q.w[4]), "=&v"(q.w[5]), "=&v"(q.w[6]), "=&v"(q.w[7]),
        "=&v"(dl), "=&v"(dh), "=&v"(sw), "=&v"(xh)
      : "v"(t.w[0]), "v"(t.w[1]), "v"(t.w[2]), "v"(t.w[3]), "v"(t.w[4]), "v"(t.w[5]), "v"(t.w[6]), "v"(t.w[7]), "s"(MP), "s"(K)
      : "vcc");
#undef ECS_K1_ROUND
#endif
  // res = T_hi + Q - E; carry of the addition and borrow of the subtraction net out to the 257th bit (the value is in [0, 2p))
  fe res;
  lane_mask top;
  asm("v_add_co_u32 %0, vcc, %9, %17\n\t"
      "v_addc_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_addc_co_u32 %7, %8, %16, %24, vcc\n\t"       /* carry -> %8 */
      "v_sub_co_u32 %0, vcc, %0, %25\n\t"
      "v_subb_co_u32 %1, vcc, %1, %26, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %2, 0, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %3, 0, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %4, 0, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %5, 0, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %6, 0, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %7, 0, vcc\n\t"
      "s_andn2_b64 %8, %8, vcc"                        /* top = carry and not borrow */
      : "=&v"(res.w[0]), "=&v"(res.w[1]), "=&v"(res.w[2]), "=&v"(res.w[3]), "=&v"(res.w[4]), "=&v"(res.w[5]), "=&v"(res.w[6]), "=&v"(res.w[7]), "=&s"(top)
      : "v"(t.w[8]), "v"(t.w[9]), "v"(t.w[10]), "v"(t.w[11]), "v"(t.w[12]), "v"(t.w[13]), "v"(t.w[14]), "v"(t.w[15]),
        "v"(q.w[0]), "v"(q.w[1]), "v"(q.w[2]), "v"(q.w[3]), "v"(q.w[4]), "v"(q.w[5]), "v"(q.w[6]), "v"(q.w[7]), "v"(dl), "v"(dh)
      : "vcc", "scc");
  cond_sub_p<CURVE_SECP256K1>(res, top);
  return res;
}

// secp256k1, classical domain: T mod p with p = 2^256 - c, c = 2^32 + 977.
//   T = H*2^256 + L  ==  L + H*c       (first fold, S < 2^289)
//   S = h2*2^256 + S_lo == S_lo + h2*c  (second fold, h2 < 2^33, result < 2^256 + 2^67)
// then the usual conditional subtraction.  8 + 2 multiplies by the 10-bit constant 977 instead of
// the 72 of a generic Montgomery reduction; H*2^32 is a word shift.
ECS_DEV fe reduce_secp256k1_classical(fe2& t) {
  const uint32_t K = 977u;
  fe lo, hi, H, A;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint64_t pj = mul_wide(t.w[8 + j], K);
    lo.w[j] = (uint32_t)pj; hi.w[j] = (uint32_t)(pj >> 32);        // hi < 2^10
    H.w[j] = t.w[8 + j]; A.w[j] = t.w[j];
  }
  const uint32_t ca = add8(A, lo);         // A = L + sum lo_j 2^(32 j)
  const uint32_t cb = add8(H, hi);         // H = H + sum hi_j 2^(32 j)     (enters one word higher)
  // S = A + (H << 32) + ca*2^256 + cb*2^288: words s0 = A0, s_j = A_j + H_(j-1), s8 = ca + H_7, s9 = cb
  uint32_t s8 = ca, s9 = cb;
  asm("v_add_co_u32 %0, vcc, %0, %9\n\t"
      "v_addc_co_u32 %1, vcc, %1, %10, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, %11, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %12, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %4, %13, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %14, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %6, %15, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %7, %16, vcc\n\t"
      "v_addc_co_u32 %8, vcc, 0, %8, vcc"
      : "+v"(A.w[1]), "+v"(A.w[2]), "+v"(A.w[3]), "+v"(A.w[4]), "+v"(A.w[5]), "+v"(A.w[6]), "+v"(A.w[7]), "+v"(s8), "+v"(s9)
      : "v"(H.w[0]), "v"(H.w[1]), "v"(H.w[2]), "v"(H.w[3]), "v"(H.w[4]), "v"(H.w[5]), "v"(H.w[6]), "v"(H.w[7])
      : "vcc");
  // second fold: F = h2*977 + (h2 << 32) with h2 = s8 + s9*2^32 (s9 <= 2), three words
  const uint64_t m0 = mul_wide(s8, K);
  uint64_t m1 = (uint64_t)(uint32_t)(m0 >> 32);
  mac_nocarry(m1, s9, K);
  uint32_t f0 = (uint32_t)m0, f1 = (uint32_t)m1, f2 = (uint32_t)(m1 >> 32);
  lane_mask top;
  asm("v_add_co_u32 %1, vcc, %1, %12\n\t"          // f1 += s8
      "v_addc_co_u32 %2, vcc, %2, %13, vcc\n\t"    // f2 += s9 + carry
      "v_add_co_u32 %3, vcc, %3, %0\n\t"           // A0 += f0
      "v_addc_co_u32 %4, vcc, %4, %1, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %2, vcc\n\t"
      "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
      "v_addc_co_u32 %7, vcc, 0, %7, vcc\n\t"
      "v_addc_co_u32 %8, vcc, 0, %8, vcc\n\t"
      "v_addc_co_u32 %9, vcc, 0, %9, vcc\n\t"
      "v_addc_co_u32 %10, %11, 0, %10, vcc"
      : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(A.w[0]), "+v"(A.w[1]), "+v"(A.w[2]), "+v"(A.w[3]), "+v"(A.w[4]), "+v"(A.w[5]), "+v"(A.w[6]), "+v"(A.w[7]), "=&s"(top)
      : "v"(s8), "v"(s9)
      : "vcc");
  cond_sub_p<CURVE_SECP256K1_CLASSICAL>(A, top);
  return A;
}

template <int CURVE> ECS_DEV fe mgry_reduce(fe2& t) {
  if constexpr (curve_prime<CURVE>::is_p256) return mgry_reduce_p256(t);
  else if constexpr (CURVE == CURVE_SECP256K1_CLASSICAL) return reduce_secp256k1_classical(t);
  else if constexpr (CURVE == CURVE_SECP256K1 || CURVE == CURVE_SECP256K1_REFSQR) return mgry_reduce_secp256k1(t);
  else return mgry_reduce_generic<CURVE>(t);
}

// a*b*R^-1 mod p                                                           mgry_ops.h:31-35
template <int CURVE> ECS_DEV fe fe_mul(const fe& a, const fe& b) {
  fe2 t = mul8x8(a, b);
  return mgry_reduce<CURVE>(t);
}
// a*a*R^-1 mod p                                                           mgry_ops.h:37-42
template <int CURVE> ECS_DEV fe fe_sqr(const fe& a) {
  fe2 t;
#if defined(ECS_SQR8_REF_C) && ECS_SQR8_REF_C
  if constexpr (curve_prime<CURVE>::ref_square) t = sqr8_ref_c(a); else t = sqr8(a);     // A/B build: round 2's restatement in C
#else
  if constexpr (curve_prime<CURVE>::ref_square) t = sqr8_ref(a); else t = sqr8(a);
#endif
  return mgry_reduce<CURVE>(t);
}
// (a*b - T) * R^-1 mod p for an UNREDUCED 512-bit product T < p^2: one reduction for a difference of two products.
// The 512-bit difference is made non-negative by adding p * 2^256 -- p into the high half, under EXEC = the lanes that
// borrowed -- so the reduction sees a value in [0, p * 2^256) and returns the canonical residue of the same field
// element fe_sub(fe_mul(a, b), reduce(T)) would: one Montgomery reduction (62 instructions) traded for 8.
template <int CURVE> ECS_DEV fe fe_mul_sub_product(const fe& a, const fe& b, const fe2& T) {
  fe2 v = mul8x8(a, b);
  lane_mask save;
#define ECS_V16 "+v"(v.w[0]), "+v"(v.w[1]), "+v"(v.w[2]), "+v"(v.w[3]), "+v"(v.w[4]), "+v"(v.w[5]), "+v"(v.w[6]), "+v"(v.w[7]), \
                "+v"(v.w[8]), "+v"(v.w[9]), "+v"(v.w[10]), "+v"(v.w[11]), "+v"(v.w[12]), "+v"(v.w[13]), "+v"(v.w[14]), "+v"(v.w[15]), "=&s"(save)
#define ECS_T16 "v"(T.w[0]), "v"(T.w[1]), "v"(T.w[2]), "v"(T.w[3]), "v"(T.w[4]), "v"(T.w[5]), "v"(T.w[6]), "v"(T.w[7]), \
                "v"(T.w[8]), "v"(T.w[9]), "v"(T.w[10]), "v"(T.w[11]), "v"(T.w[12]), "v"(T.w[13]), "v"(T.w[14]), "v"(T.w[15])
#define ECS_SUB16 \
      "v_sub_co_u32 %0, vcc, %0, %17\n\t"   "v_subb_co_u32 %1, vcc, %1, %18, vcc\n\t"  "v_subb_co_u32 %2, vcc, %2, %19, vcc\n\t"  "v_subb_co_u32 %3, vcc, %3, %20, vcc\n\t" \
      "v_subb_co_u32 %4, vcc, %4, %21, vcc\n\t"  "v_subb_co_u32 %5, vcc, %5, %22, vcc\n\t"  "v_subb_co_u32 %6, vcc, %6, %23, vcc\n\t"  "v_subb_co_u32 %7, vcc, %7, %24, vcc\n\t" \
      "v_subb_co_u32 %8, vcc, %8, %25, vcc\n\t"  "v_subb_co_u32 %9, vcc, %9, %26, vcc\n\t"  "v_subb_co_u32 %10, vcc, %10, %27, vcc\n\t" "v_subb_co_u32 %11, vcc, %11, %28, vcc\n\t" \
      "v_subb_co_u32 %12, vcc, %12, %29, vcc\n\t" "v_subb_co_u32 %13, vcc, %13, %30, vcc\n\t" "v_subb_co_u32 %14, vcc, %14, %31, vcc\n\t" "v_subb_co_u32 %15, vcc, %15, %32, vcc\n\t" \
      "s_and_saveexec_b64 %16, vcc\n\t"
  if constexpr (curve_prime<CURVE>::is_p256) {
    asm(ECS_SUB16
        "v_add_co_u32 %8, vcc, -1, %8\n\t"   "v_addc_co_u32 %9, vcc, -1, %9, vcc\n\t"   "v_addc_co_u32 %10, vcc, -1, %10, vcc\n\t" "v_addc_co_u32 %11, vcc, 0, %11, vcc\n\t"
        "v_addc_co_u32 %12, vcc, 0, %12, vcc\n\t" "v_addc_co_u32 %13, vcc, 0, %13, vcc\n\t" "v_addc_co_u32 %14, vcc, 1, %14, vcc\n\t" "v_addc_co_u32 %15, vcc, -1, %15, vcc\n\t"
        "s_mov_b64 exec, %16"
        : ECS_V16 : ECS_T16 : "vcc", "scc");
  } else {
    using K = curve_consts<CURVE>;
    const uint32_t p0 = K::P[0], p1 = K::P[1];
    asm(ECS_SUB16
        "v_add_co_u32 %8, vcc, %8, %33\n\t"  "v_addc_co_u32 %9, vcc, %9, %34, vcc\n\t"  "v_addc_co_u32 %10, vcc, -1, %10, vcc\n\t" "v_addc_co_u32 %11, vcc, -1, %11, vcc\n\t"
        "v_addc_co_u32 %12, vcc, -1, %12, vcc\n\t" "v_addc_co_u32 %13, vcc, -1, %13, vcc\n\t" "v_addc_co_u32 %14, vcc, -1, %14, vcc\n\t" "v_addc_co_u32 %15, vcc, -1, %15, vcc\n\t"
        "s_mov_b64 exec, %16"
        : ECS_V16 : ECS_T16, "v"(p0), "v"(p1) : "vcc", "scc");
  }
#undef ECS_SUB16
#undef ECS_T16
#undef ECS_V16
  return mgry_reduce<CURVE>(v);
}

// n*R mod p = mgry_reduce(n * (R^2 mod p))                                 mgry.h:47-50
template <int CURVE> ECS_DEV fe fe_from_classical(const fe& n) {
  if constexpr (CURVE == CURVE_SECP256K1 || CURVE == CURVE_SECP256K1_REFSQR) {
    // R = 2^256 = 2^32 + 977 (mod p): n*R mod p is the pseudo-Mersenne fold of n * 2^256 -- the classical reduction applied to
    // the 512-bit value (hi = n, lo = 0).  Same canonical residue as the Montgomery route, 61 instructions instead of 373.
    fe2 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) { t.w[i] = 0; t.w[8 + i] = n.w[i]; }
    return reduce_secp256k1_classical(t);
  } else {
    return fe_mul<CURVE>(n, FE_CONST(CURVE, RSQ));
  }
}
// n*R^-1 mod p = mgry_reduce(zero-extended n)                              mgry.h:52-55
template <int CURVE> ECS_DEV fe fe_to_classical(const fe& n) {
  fe2 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) { t.w[i] = n.w[i]; t.w[8 + i] = 0; }
  return mgry_reduce<CURVE>(t);
}

// ---------------------------------------------------------------- masked select / swap
// m is an all-ones / all-zeros word per lane (the reference's lane mask, utility.h:45-51).
#ifndef ECS_CSWAP_CNDMASK
#define ECS_CSWAP_CNDMASK 0       // 1: one v_cmp + 16 v_cndmask by an SGPR mask instead of the XOR / AND form (the compiler lowers that to 9 v_xor +
                                  // 16 v_bitop3 per ladder iteration).  Measured (profiles/r03/ab_cswap_cndmask.txt): see DESIGN.md section 9
#endif
ECS_DEV void fe_cswap(uint32_t m, fe& a, fe& b) {                            // swap.h:15-22
#if ECS_CSWAP_CNDMASK
  lane_mask k;
  asm("v_cmp_ne_u32_e64 %0, %1, 0" : "=s"(k) : "v"(m));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t na, nb;
    asm("v_cndmask_b32_e64 %0, %2, %3, %4\n\tv_cndmask_b32_e64 %1, %3, %2, %4" : "=&v"(na), "=&v"(nb) : "v"(a.w[i]), "v"(b.w[i]), "s"(k));
    a.w[i] = na; b.w[i] = nb;
  }
#else
#pragma unroll
  for (int i = 0; i < 8; ++i) { uint32_t t = (a.w[i] ^ b.w[i]) & m; a.w[i] ^= t; b.w[i] ^= t; }
#endif
}
ECS_DEV fe fe_select(uint32_t m, const fe& a, const fe& b) {                 // ifelse.h:15-22 (m ? a : b)
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = b.w[i] ^ ((a.w[i] ^ b.w[i]) & m);
  return r;
}
ECS_DEV bool fe_eq(const fe& a, const fe& b) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d |= a.w[i] ^ b.w[i];
  return d == 0;
}

// a^e for a public, wave-uniform exponent e (256 bits, LE words), LSB first.
// Same multiplication sequence as mgry_ops.h:44-86 (result *= base when the bit is set, base is
// squared between bits, squaring stops after the top set bit).
template <int CURVE> ECS_DEV fe fe_pow(const fe& a, const uint32_t (&e)[8]) {
  fe result = FE_CONST(CURVE, R_P);
  int top = -1;
  for (int i = 255; i >= 0; --i) if ((e[i >> 5] >> (i & 31)) & 1u) { top = i; break; }
  fe base = a;
  for (int i = 0; i <= top; ++i) {
    if ((e[i >> 5] >> (i & 31)) & 1u) result = fe_mul<CURVE>(result, base);
    if (i < top) base = fe_sqr<CURVE>(base);
  }
  return result;
}

}  // namespace ecsimd_hip
