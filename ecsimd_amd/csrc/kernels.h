// kernels.h -- host-side launchers, one per kernel family (defined in the k_*.hip files, which are
// compiled in parallel by the Makefile; the ladder alone takes ~1 min of hipcc time per curve).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ecsimd_hip {
struct gmod;                 // gfield.cuh: a run-time modulus and what the field layer derives from it
struct gcurve;               // gcurve.cuh: a run-time curve (its field, a R, b R, the generator, the ladder loop's 29-bit constants)
namespace launch {

constexpr int BLOCK = 256;   // one wave per SIMD of a CU; several workgroups resident per CU
inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

struct words8 { uint32_t w[8]; };   // a 256-bit kernel argument (exponent / shared scalar)

// k_bignum.hip (curve independent)
void add(hipStream_t, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* carry, size_t n);
void sub(hipStream_t, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* borrow, size_t n);
void sub_if_above(hipStream_t, const uint64_t* a, const uint64_t* p, uint64_t* out, size_t n);
void shift_left_one(hipStream_t, const uint64_t* a, uint64_t* out, uint8_t* carry, size_t n);
void mul(hipStream_t, const uint64_t* a, const uint64_t* b, uint64_t* out8, size_t n);
void square(hipStream_t, const uint64_t* a, uint64_t* out8, size_t n, bool ref_compat = false);   // ref_compat: mul.h:160-212 as written
void swap_if(hipStream_t, const uint8_t* mask, uint64_t* a, uint64_t* b, size_t n);
void if_else(hipStream_t, const uint8_t* mask, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void patch_special(hipStream_t, const uint64_t* k, const uint64_t* special /* 3 scalars, 3 x, 3 y */, uint64_t* ox, uint64_t* oy, size_t n);   // oy may be null
void cmp_eq(hipStream_t, const uint64_t* a, const uint64_t* b, int limbs, uint8_t* flag, size_t n);
void mask_op(hipStream_t, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);     // 0 not, 1 and, 2 or, 3 equal
void mask_count(hipStream_t, const uint8_t* a, size_t n, unsigned long long* count);
void fill_random(hipStream_t, uint64_t* out, size_t n, uint64_t seed, uint64_t stream, uint64_t first, int clear_top);
void peak_mad32(hipStream_t, int blocks, uint32_t* sink, int iters, uint32_t seed);
constexpr int PEAK_MADS_PER_LANE_PER_ITER = 64;

// k_serial.hip (wire formats; HBM-bound)
void bytes_be(hipStream_t, const void* in, void* out, size_t n);
void mask_bit(hipStream_t, const uint64_t* a, int bit, uint8_t* flag, size_t n);
void wide4_to_lanes(hipStream_t, const void* wides, size_t record_bytes, size_t offset_bytes, uint64_t* out, size_t n);      // n = ELEMENTS (4 per wide)
void lanes_to_wide4(hipStream_t, const uint64_t* in, void* wides, size_t record_bytes, size_t offset_bytes, size_t n);
void sec1_encode(hipStream_t, int curve, const uint64_t* x, const uint64_t* y, uint8_t* out, size_t n, bool compressed);
void sec1_decode(hipStream_t, int curve, const uint8_t* in, uint64_t* x, uint64_t* y, uint8_t* ok, size_t n, bool compressed);
void on_curve(hipStream_t, int curve, const uint64_t* x, const uint64_t* y, uint8_t* ok, size_t n);                 // classical (x, y): x, y < p and on the curve
void clear_invalid(hipStream_t, const uint8_t* valid, uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n);      // (0, 0) / not finite where !valid

// k_field.hip
enum field_op { F_MOD_ADD, F_MOD_SUB, F_MGRY_MUL, F_MGRY_SQR, F_FROM_CLASSICAL, F_TO_CLASSICAL, F_INVERSE, F_OPPOSITE };
void field_binop(hipStream_t, int curve, field_op op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void field_unop(hipStream_t, int curve, field_op op, const uint64_t* a, uint64_t* out, size_t n);
void mod_shift_left(hipStream_t, int curve, const uint64_t* a, int count, uint64_t* out, size_t n);
void mod_mul(hipStream_t, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void mgry_reduce(hipStream_t, int curve, const uint64_t* a8, uint64_t* out, size_t n);
void mgry_pow(hipStream_t, int curve, const uint64_t* a, const words8& e, uint64_t* out, size_t n);
void gfp_sqrt(hipStream_t, int curve, const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n);

// k_gfield.hip: the same layer for a RUN-TIME modulus (any odd 256-bit value: the group orders, a caller's own prime); ref_square =
// the reference's square() as written.  F_INVERSE: division steps for a prime modulus, x^(p-2) bit by bit otherwise (gfp.h:42-44).
void gfield_binop(hipStream_t, const gmod&, field_op op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void gfield_unop(hipStream_t, const gmod&, field_op op, const uint64_t* a, uint64_t* out, size_t n, bool ref_square);
void gfield_mod_mul(hipStream_t, const gmod&, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void gfield_shift_left(hipStream_t, const gmod&, const uint64_t* a, int count, uint64_t* out, size_t n);
void gfield_reduce(hipStream_t, const gmod&, const uint64_t* a8, uint64_t* out, size_t n);
void gfield_pow(hipStream_t, const gmod&, const uint64_t* a, const words8& e, uint64_t* out, size_t n, bool ref_square);
void gfield_sqrt(hipStream_t, const gmod&, const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n, bool ref_square);
void gfield_inverse_batched(hipStream_t, const gmod&, const uint64_t* a, uint64_t* out, size_t n);   // prime modulus; out must not alias a
// ECDSA verification's arithmetic modulo the group order (gmod of n): valid = 1 <= r, s < n; u1 = e / s, u2 = r / s (0, 0 where invalid)
void ecdsa_scalars(hipStream_t, const gmod& order, const uint64_t* e, const uint64_t* r, const uint64_t* s, uint64_t* u1, uint64_t* u2, uint8_t* valid, size_t n);
// ECDSA signing's arithmetic modulo the group order: r = x mod n, s = (e + r d) / k; ok = the inputs are in range and r, s != 0 (secret d, k: selects only)
void ecdsa_sign_scalars(hipStream_t, const gmod& order, const uint64_t* e, const uint64_t* d, const uint64_t* k, const uint64_t* x, uint64_t* r, uint64_t* s, uint8_t* ok, size_t n);

// k_gcurve.hip / k_gladder.hip: the point layer and the ladder for a curve registered at RUN time (curve_group<Curve> for any Curve: curve.h:12-15).
// ref = the reference's square() as written; gc_scalar_mult flags: ECSIMD_HIP_BASE_MGRY | LADDER_RADIX32 | REF_SQUARE_COMPAT (Jacobian Montgomery out).
void gc_from_affine(hipStream_t, const gcurve&, const uint64_t* x, const uint64_t* y, uint64_t* jx, uint64_t* jy, uint64_t* jz, size_t n);
void gc_to_affine(hipStream_t, const gcurve&, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n, bool ref);
void gc_to_affine_batched(hipStream_t, const gcurve&, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n);   // x / y must not alias the inputs
void gc_compute_y(hipStream_t, const gcurve&, const uint64_t* x, uint64_t* y, uint8_t* ok, size_t n, bool ref);
void gc_on_curve(hipStream_t, const gcurve&, const uint64_t* x, const uint64_t* y, uint8_t* ok, size_t n);
void gc_dblu(hipStream_t, const gcurve&, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref);
void gc_zaddu(hipStream_t, const gcurve&, uint64_t* px, uint64_t* py, uint64_t* pz, const uint64_t* qx, const uint64_t* qy, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref);
void gc_zdau(hipStream_t, const gcurve&, const uint64_t* px, const uint64_t* py, const uint64_t* pz, uint64_t* qx, uint64_t* qy, uint64_t* qz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref);
void gc_add_z2_1(hipStream_t, const gcurve&, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref);
void gc_trplu(hipStream_t, const gcurve&, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n, bool ref);
void gc_scalar_mult(hipStream_t, const gcurve&, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags);
// k_gcomb.hip: k G on a registered curve from a 4-bit odd-digit table of multiples of its generator in LDS (64 windows x 8 entries x 64 B + k* G + k*)
constexpr int GCOMB_WINDOWS = 64, GCOMB_ENTRIES = 8;
void gc_pack_table(hipStream_t, const gcurve& G, const uint64_t* tx, const uint64_t* ty, uint32_t* table, int entries);
void gc_base_windowed(hipStream_t, const gcurve& G, const words8& order, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, bool constant_time);
// ... and from signed 7-bit windows with odd digits (37 windows x 64 entries, 148 KiB of LDS, summed from the bottom: ALG_WINDOWED_SIGNED; public scalars)
// ... and the constant-time 5-bit comb (52 windows x 16 entries, 53 KB of LDS, every entry of a window read: ALG_WINDOWED | ALG_CONSTANT_TIME, k G of ecdsa_sign)
constexpr int GCOMB7_BITS = 7, GCOMB7_WINDOWS = 37, GCOMB7_ENTRIES = 64, GCOMB5_WINDOWS = 52, GCOMB5_ENTRIES = 16, GCOMB20_WINDOWS = 13, GCOMB20_ENTRIES = 1 << 19;   // (20 bits: 436 MB in device memory, ALG_WINDOWED_BIG)
void gc_base_windowed_s(hipStream_t, const gcurve& G, const words8& order, int bits, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n);
// k_gvarwin.hip: k P on a registered curve with per-lane window tables (the eight odd multiples of P over one Z, the loop on the isomorphic curve; affine
// classical out, oy may be null).  scratch: gc_varwin_scratch_bytes(n) bytes, 32-byte aligned; k_stride, x, y as for gc_scalar_mult; flags: ECSIMD_HIP_BASE_MGRY.
inline size_t gc_varwin_scratch_bytes(size_t n) { return n * (8 * 64 + 32 + 3 * 32); }
void gc_varwin_scalar_mult(hipStream_t, const gcurve& G, const words8& order, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, int flags,
                           uint64_t* scratch, uint64_t* ox, uint64_t* oy, size_t n);
// ... and what ECDSA on a registered curve needs on top (public-data affine addition, the acceptance test, the ladder's three degenerate scalars worked around)
void gc_affine_add_batched(hipStream_t, const gcurve&, const uint64_t* ax, const uint64_t* ay, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n);
void gc_x_mod_n_equals(hipStream_t, const gmod& order, const uint64_t* x, const uint8_t* finite, const uint64_t* r, uint8_t* ok, size_t n);
void gc_ladder_safe_scalars(hipStream_t, const gmod& order, const uint64_t* u, uint64_t* adj, uint8_t* neg, size_t n);
void gc_negate_where(hipStream_t, const gcurve&, const uint8_t* neg, uint64_t* y, size_t n);
void gc_sec1_decode(hipStream_t, const gcurve&, const uint8_t* in, uint64_t* x, uint64_t* y, uint8_t* ok, size_t n, bool compressed);
void gc_zdau_repeat(hipStream_t, const gcurve&, const uint64_t* px, const uint64_t* py, const uint64_t* pz, const uint64_t* qx, const uint64_t* qy,
                    uint64_t* rx, uint64_t* ry, uint64_t* sx, uint64_t* sy, uint64_t* oz, size_t n, int iters, uint64_t swap_bits, int radix);

// k_fe29_raw.hip: one function of fe29.cuh on raw 9-limb operands (the diagnostic entry ecsimd_hip_fe29_raw)
enum fe29_raw_op { RAW_ZDAU = 0, RAW_MADD = 1, RAW_JDBL = 2, RAW_DBL_ADD = 3, RAW_MADDV = 4, RAW_PDBL = 5, RAW_PADD = 6, RAW_MUL = 7, RAW_SQR = 8, RAW_GJDBL = 9, RAW_ZADDU = 10 };
constexpr int fe29_raw_inputs(int op) { return op == RAW_ZDAU ? 6 : op == RAW_MUL ? 2 : op == RAW_SQR ? 1 : (op == RAW_JDBL || op == RAW_PDBL) ? 3 : op == RAW_GJDBL ? 4 : 5; }
constexpr int fe29_raw_outputs(int op) { return (op == RAW_ZDAU || op == RAW_ZADDU) ? 6 : (op == RAW_MUL || op == RAW_SQR) ? 1 : op == RAW_GJDBL ? 4 : 3; }
bool fe29_raw(hipStream_t, int curve, const gcurve* G, int op, const int32_t* in, int32_t* out, size_t n, uint32_t swap);

// k_point_<curve>.hip
void from_affine(hipStream_t, int curve, const uint64_t* x, const uint64_t* y, uint64_t* jx, uint64_t* jy, uint64_t* jz, size_t n);
void to_affine(hipStream_t, int curve, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n);
void compute_y(hipStream_t, int curve, const uint64_t* x, uint64_t* y, uint8_t* ok, size_t n);
void dblu(hipStream_t, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
void zaddu(hipStream_t, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, const uint64_t* qx, const uint64_t* qy, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
void zdau(hipStream_t, int curve, const uint64_t* px, const uint64_t* py, const uint64_t* pz, uint64_t* qx, uint64_t* qy, uint64_t* qz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
void add_z2_1(hipStream_t, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
void trplu(hipStream_t, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);

// k_ladder_<curve>.hip.  k_stride = 4 (u64 per element) for per-element scalars, 0 for one shared
// scalar (device memory either way).  x == nullptr selects the curve generator as base point.
// flags: ECSIMD_HIP_BASE_* | ECSIMD_HIP_OUT_*.
void scalar_mult(hipStream_t, int curve, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y,
                 uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags);
void zdau_repeat(hipStream_t, int curve, const uint64_t* px, const uint64_t* py, const uint64_t* pz, const uint64_t* qx, const uint64_t* qy,
                 uint64_t* rx, uint64_t* ry, uint64_t* sx, uint64_t* sy, uint64_t* oz, size_t n, int iters, uint64_t swap_bits, int radix);
// k_affine_<curve>.hip: simultaneous-inversion to_affine (x, y must not alias the inputs), the
// 4-bit-window table packer and the fixed-base windowed multiplication (Jacobian out, fast domain).
void to_affine_batched(hipStream_t, int curve, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n, bool in_fast_domain);
void inverse_batched(hipStream_t, int curve, const uint64_t* a, uint64_t* out, size_t n);     // out must not alias a
void x_mod_n_equals(hipStream_t, int curve, const uint64_t* x, const uint8_t* finite, const uint64_t* r, uint8_t* ok, size_t n);
void affine_add_batched(hipStream_t, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n);
void pack_table(hipStream_t, int curve, const uint64_t* tx, const uint64_t* ty, uint32_t* table);
void base_windowed(hipStream_t, int curve, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, bool constant_time);
// signed windows of wbits = 6 or 7 bits
void pack_table_signed(hipStream_t, int curve, int wbits, const uint64_t* tx, const uint64_t* ty, uint32_t* table);
void base_windowed_signed(hipStream_t, int curve, int wbits, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, bool constant_time);

// k_varwin_<curve>.hip: variable-base multiplication with per-lane window tables of 8 multiples of P (affine out, classical).
// scratch: varwin_scratch_bytes(n) bytes, 32-byte aligned; k_stride, x, y as for scalar_mult (flags: ECSIMD_HIP_BASE_*).
void varwin_scalar_mult(hipStream_t, int curve, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, int flags,
                        uint64_t* scratch, uint64_t* ox, uint64_t* oy, size_t n);
// complete mixed addition (A Jacobian, Z = 0 is infinity; B Montgomery-form affine, (0, 0) is infinity)
void add_mixed_complete(hipStream_t, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by,
                        uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
inline size_t varwin_scratch_bytes(size_t n) { return n * (7 * 4 * 32 + 8 * 64); }

// signed BIG_WINDOW_BITS-bit windows over a table in device memory (20 bits: 13 windows x 524 288 entries x 64 B = 436 MB)
void pack_table_big(hipStream_t, int curve, const uint64_t* tx, const uint64_t* ty, uint32_t* table);
void base_windowed_big(hipStream_t, int curve, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n);
#ifndef ECS_BIG_WINDOW_BITS
#define ECS_BIG_WINDOW_BITS 20   // measured on one MI355X, P-256 / secp256k1 M/s: 16 bits 904 / 883 (36 MB), 18 bits 995 / 965 (126 MB),
                                 // 20 bits 1 075 / 1 090 (436 MB, built in 0.23 s), 22 bits 1 124 / 1 146 (1.6 GB, 0.8 s)
#endif
constexpr int BIG_WINDOW_BITS = ECS_BIG_WINDOW_BITS;

// per-curve pieces (one translation unit each)
template <int C> struct point_launch {
  static void from_affine(hipStream_t, const uint64_t*, const uint64_t*, uint64_t*, uint64_t*, uint64_t*, size_t);
  static void to_affine(hipStream_t, const uint64_t*, const uint64_t*, const uint64_t*, uint64_t*, uint64_t*, size_t);
  static void compute_y(hipStream_t, const uint64_t*, uint64_t*, uint8_t*, size_t);
  static void dblu(hipStream_t, uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, size_t);
  static void zaddu(hipStream_t, uint64_t*, uint64_t*, uint64_t*, const uint64_t*, const uint64_t*, uint64_t*, uint64_t*, uint64_t*, size_t);
  static void zdau(hipStream_t, const uint64_t*, const uint64_t*, const uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, size_t);
  static void add_z2_1(hipStream_t, const uint64_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint64_t*, uint64_t*, uint64_t*, uint64_t*, size_t);
  static void trplu(hipStream_t, uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, size_t);
  static void scalar_mult(hipStream_t, const uint64_t*, int, const uint64_t*, const uint64_t*, uint64_t*, uint64_t*, uint64_t*, size_t, int);
  static void zdau_repeat(hipStream_t, const uint64_t* px, const uint64_t* py, const uint64_t* pz, const uint64_t* qx, const uint64_t* qy,
                          uint64_t* rx, uint64_t* ry, uint64_t* sx, uint64_t* sy, uint64_t* oz, size_t n, int iters, uint64_t swap_bits, int radix);
  // x(kP) only, by the ladder without Z (defined for P-256 only: needs a != 0; k_ladder.inc)
  static void scalar_mult_x(hipStream_t, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* scratch, size_t n, int flags);
  // k_affine_<curve>.hip
  static void to_affine_batched(hipStream_t, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n, bool in_fast_domain);
  static void inverse_batched(hipStream_t, const uint64_t* a, uint64_t* out, size_t n);
  static void x_mod_n_equals(hipStream_t, const uint64_t* x, const uint8_t* finite, const uint64_t* r, uint8_t* ok, size_t n);
  static void affine_add_batched(hipStream_t, const uint64_t* ax, const uint64_t* ay, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n);
  static void pack_table(hipStream_t, const uint64_t* tx, const uint64_t* ty, uint32_t* table);
  static void base_windowed(hipStream_t, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, bool constant_time);
  static void pack_table_signed(hipStream_t, int wbits, const uint64_t* tx, const uint64_t* ty, uint32_t* table);
  static void base_windowed_signed(hipStream_t, int wbits, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, bool constant_time);
  static void pack_table_big(hipStream_t, const uint64_t* tx, const uint64_t* ty, uint32_t* table);
  static void base_windowed_big(hipStream_t, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n);
  // k_varwin_<curve>.hip
  static void add_mixed_complete(hipStream_t, const uint64_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint64_t*, uint64_t*, uint64_t*, uint64_t*, size_t);
  static void varwin_scalar_mult(hipStream_t, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, int flags, uint64_t* scratch, uint64_t* ox, uint64_t* oy, size_t n);
};
inline size_t scalar_mult_x_scratch_bytes(size_t n) { return 4 * n * 32 + ((n + 31) & ~(size_t)31); }   // odd scalars, num, den, 1/den, zero flags
// The 4-bit fixed-base table in LDS (BASELINE configs[2]).  ECS_FIXED4_ODD = 1 (round 3): odd digits only -- the regular recoding of the
// odd one of k mod n, n - k, as in the big-window kernel: 64 windows x 8 odd multiples (2d + 1) 16^w G = 32 KiB, 63 mixed additions, no
// zero digit and therefore no "skip" / "infinity" selects.  0: round 1's unsigned digits (64 x 16 entries d 16^w G, 64 additions).
#ifndef ECS_FIXED4_ODD
#define ECS_FIXED4_ODD 1
#endif
constexpr int FIXED4_ENTRIES = ECS_FIXED4_ODD ? 8 : 16;
// The signed 6- / 7-bit LDS tables (ALG_WINDOWED_SIGNED).  ECS_SIGNED_ODD = 1 (round 3): odd digits as above -- 37 windows x 64 odd multiples
// (2d + 1) 2^(7w) G for 7 bits, 36 mixed additions, no carry window and no skip / infinity selects; 0: round 1's carry recoding (digits in [-63, 64]).
#ifndef ECS_SIGNED_ODD
#define ECS_SIGNED_ODD 1
#endif
constexpr int signed_windows(int bits) { return ECS_SIGNED_ODD ? (256 + bits - 1) / bits : (256 + bits) / bits; }
constexpr size_t WINDOW_TABLE_BYTES = 64 * FIXED4_ENTRIES * 64;   // 64 windows x entries x (x, y)

}  // namespace launch
}  // namespace ecsimd_hip
