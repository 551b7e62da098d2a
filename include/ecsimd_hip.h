/*
 * ecsimd_hip.h -- C ABI of the MI355X (gfx950) batched elliptic-curve engine.
 *
 * This is the drop-in boundary for the hot path of aguinet/ecsimd.  The
 * reference is header-only C++ with no ABI of its own: its single linkable symbol is
 * scalar_mult_p256 (lib/scalar_mult_p256.cpp:10-12).  Every entry point below replaces one
 * reference function (cited per declaration) over a RUNTIME-LENGTH batch; n = 4 reproduces one
 * eve::wide of the reference.  The C++ headers in include/ecsimd/ bind these symbols and keep
 * the reference's names (curve_group<Curve>::scalar_mult, DBLU, ZADDU, ZDAU, ADD_Z2_1,
 * scalar_mult_p256, ...).  INTEGRATION.md shows the binding a maintainer of the reference adds.
 *
 * Data layout (all pointers are DEVICE pointers, 16-byte aligned):
 *   field element / scalar : 4 x uint64_t, little-endian limb order (limb 0 least significant),
 *                            = the reference's bignum<uint64_t,4> (bignum.h:38-99,
 *                            serialization.h:18-21).  A batch is AoS: element i at p + 4*i.
 *   512-bit product        : 8 x uint64_t per element.
 *   point batch            : one array per coordinate (x[], y[], z[]).  Jacobian coordinates are
 *                            in Montgomery form (jacobian_curve_point.h:25-31), affine ones are
 *                            classical unless the function name says _mgry.
 *   flags                  : one uint8_t per element (0/1).
 * Preconditions the reference only asserts in debug builds (co-Z inputs, Z = mgry(1) for
 * DBLU / ADD_Z2_1 / scalar_mult) are the caller's responsibility here too.
 *
 * Every call enqueues kernels on the context's stream and returns without synchronising;
 * ecsimd_hip_sync() waits.  Return value: 0 on success, negative ecsimd_hip_status otherwise.
 * Nothing throws across this boundary.  A context is not thread-safe: one per host thread/GPU.
 */
#ifndef ECSIMD_HIP_H
#define ECSIMD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ecsimd_hip_ctx ecsimd_hip_ctx;

enum ecsimd_hip_curve {
  ECSIMD_HIP_P256 = 0,      /* curve_nist_p256.h:14-32 */
  ECSIMD_HIP_SECP256K1 = 1, /* prime of tests/mgry.cpp:25-27 with a=0, b=7 (SEC 2) */
  ECSIMD_HIP_FIRST_REGISTERED_CURVE = 0x10000  /* ids of ecsimd_hip_register_curve start here (r5): curve_group<Curve> for ANY Curve type, curve.h:12-15 */
};

/* FIELD ids: what the element-wise field entry points (mod_*, mgry_*, gfp_*, get_constant) take as `curve`.  The reference's field layer is
 * generic in the modulus type P (mgry_mul.h:84-121 details::mgry_reduce<P>, mgry_csts.h:15-35 mgry_constants<WBN, P>, gfp.h:17-115 GFp<WBN, P>);
 * here a modulus is a run-time value with an id.  0, 1: the curve primes above (hand-laid special-form kernels); 2, 3: the two group orders n
 * (what ECDSA computes modulo); 4...: moduli registered with ecsimd_hip_register_modulus; a curve id of ecsimd_hip_register_curve names its prime's field. */
enum ecsimd_hip_field {
  ECSIMD_HIP_FIELD_P256_ORDER = 2,      /* n of P-256 (SP 800-186 3.2.1.3) */
  ECSIMD_HIP_FIELD_SECP256K1_ORDER = 3  /* n of secp256k1 (SEC 2 v2 2.4.1) */
};
enum { ECSIMD_HIP_MODULUS_PRIME = 1 };  /* register_modulus: the caller vouches that p is PRIME; gfp_inverse then uses the constant-time division-step
                                           inversion with one shared inversion per up to 128 elements (the unique inverse = x^(p-2)).  Without it
                                           gfp_inverse raises to p - 2 bit by bit -- gfp.h:42-44 as written, the reference's value for ANY odd p */

enum ecsimd_hip_status {
  ECSIMD_HIP_OK = 0,
  ECSIMD_HIP_ERR_BAD_ARG = -1,     /* null pointer, unknown curve, misaligned pointer */
  ECSIMD_HIP_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture (gfx950 code object only) */
  ECSIMD_HIP_ERR_HIP = -3          /* a HIP runtime call failed; see ecsimd_hip_last_error() */
};

/* scalar_mult flags.  SECRET SCALARS: only the default algorithm (no ALG_* bit: the reference's co-Z ladder, and its Z-less
 * form for x-only P-256 output) is constant-time -- one instruction stream, no branch or address that depends on a scalar
 * bit or an intermediate value, checked on the shipped ISA by tests/test_constant_time_isa.py.  Every ALG_* bit selects a
 * table-driven algorithm whose memory addresses are scalar digits: for PUBLIC scalars only (each flag says so below). */
enum {
  ECSIMD_HIP_BASE_CLASSICAL = 0,   /* base point (x, y) classical: from_affine is applied first */
  ECSIMD_HIP_BASE_MGRY = 1,        /* base point already Montgomery form (what scalar_mult_p256 receives) */
  ECSIMD_HIP_BASE_GENERATOR = 512, /* ecsimd_hip_scalar_mult only, with x = y = NULL: the base point is the curve generator (= ecsimd_hip_scalar_mult_base).
                                      Two null pointers WITHOUT this flag are ECSIMD_HIP_ERR_BAD_ARG, like any other null pointer */
  ECSIMD_HIP_OUT_JACOBIAN = 0,     /* out = (X, Y, Z) Montgomery form, as curve_group::scalar_mult returns */
  ECSIMD_HIP_OUT_AFFINE = 2,       /* out = to_affine(): (x, y) classical; oz may be NULL; oy may be NULL too: the x coordinate only
                                      (ECDH's shared secret, ECDSA's r): the conversion then skips y's two field multiplications, and on P-256 the
                                      ladder itself (no ALG_* flag, no REF_SQUARE_COMPAT) runs WITHOUT the Z coordinate -- 8M + 6S per bit instead
                                      of 9M + 7S, x recovered from the two co-Z results and the curve equation, 55 M/s against 48 -- on the odd one
                                      of k mod n and n - k mod n.  Same constant-time shape as the ladder; correct for EVERY 256-bit k (x = 0 for
                                      k = 0 mod n), including the reference ladder's degenerate scalars */
  ECSIMD_HIP_ALG_WINDOWED = 4,     /* with OUT_AFFINE only.  scalar_mult_base: 4-bit windows over an LDS-resident table (32 KiB: the odd
                                      multiples (2d+1)*16^w*G; the odd one of k mod n, n - k is recoded into 64 odd digits, so the sum is
                                      63 mixed additions with no zero digit to skip) and one simultaneous inversion instead of the
                                      reference's ladder.
                                      scalar_mult / double_scalar_mult (variable base): a per-element table of 8 multiples
                                      of P in device memory and signed 4-bit windows (odd digits, 3 doublings + one fused
                                      double-add per window: ~2 770 field multiplications against the ladder's 4 064;
                                      secp256k1 splits k = k1 + k2*lambda first), 1 408 B of context workspace per
                                      element, at most 2^22 elements at a time.  Same affine result as
                                      the ladder for every k except the ladder's degenerate scalars k = n-1, 2^256-n-1,
                                      2^256-n (there the reference returns a meaningless point, these paths the right
                                      one); k = 0 mod n -> (0, 0).  NOT for secret scalars: table reads (LDS for the fixed
                                      base, device memory for the per-element tables) are indexed by scalar digits --
                                      unless ALG_CONSTANT_TIME is added (below) */
  ECSIMD_HIP_ALG_WINDOWED_SIGNED = 8, /* as ALG_WINDOWED with signed 7-bit windows: 36 mixed additions instead of 63, a 148 KiB
                                      table of the odd multiples (2d+1)*2^(7i)*G (d = 0..63) in LDS, negative digits negate y; same results.
                                      NOT for secret scalars (LDS reads indexed by scalar digits) */
  ECSIMD_HIP_ALG_NO_ENDOMORPHISM = 16, /* secp256k1 + ALG_WINDOWED on a variable base splits k = k1 + k2*lambda (GLV) and runs
                                      half as many windows; this flag keeps the plain odd-digit loop of 63 windows (same results);
                                      a modifier of ALG_WINDOWED: NOT for secret scalars either way */
  ECSIMD_HIP_REF_SQUARE_COMPAT = 64, /* ladder only: square with the reference's square() AS WRITTEN (mul.h:160-212), which drops a
                                      carry at mul.h:186-190 (its "TODO: carry?", mul.h:207) on ~2e-9 of random operands -- ~3e-6 of
                                      random scalar multiplications then differ from the exact result.  With this flag (or the
                                      context option below) the output is the reference's bits on EVERY input, at ~0.9x the speed;
                                      without it, it is the exact k*P (what the reference's own tests assume).  Same ladder, same
                                      constant-time shape: safe for secret scalars */
  ECSIMD_HIP_ALG_CONSTANT_TIME = 128, /* a modifier of ALG_WINDOWED (r3): the table-driven algorithms WITHOUT a scalar-dependent address or branch -- every
                                      lane reads ALL entries a window could pick and keeps its own under lane masks; the recodings have no zero digit,
                                      exceptional scalars and k = 0 mod n are handled by selects (tools/ct_check.py checks the shipped ISA of the window
                                      loops, tests/test_constant_time_isa.py).  SAFE for secret scalars; same results as ALG_WINDOWED.
                                      scalar_mult_base: an odd-digit comb over an LDS table -- one address per wave, an LDS broadcast (52 five-bit
                                      windows x 16 entries, 51 additions, three 256-thread workgroups per CU): k*G for key generation and ECDSA
                                      nonces at 6.3x the ladder's rate (P-256 368 M/s, secp256k1 365 M/s).
                                      scalar_mult / scalar_mult_1s: the per-element window tables with all 8 entries of the lane's own table (512
                                      contiguous bytes) read in every window; on secp256k1 the GLV split stays, run on the COMPLETE addition law of
                                      a = 0 curves (no exceptional case to branch on; ALG_NO_ENDOMORPHISM: the plain odd-digit loop).  ECDH with a
                                      secret scalar at 1.33x (P-256: 78.2 M/s) / 1.67x (secp256k1: 99.8 M/s) the round-4 ladder's rate
                                      (oy = NULL works here too).  Not with ALG_WINDOWED_SIGNED / ALG_WINDOWED_BIG (64 or 2^19 entries per window to read).
                                      Without ALG_WINDOWED the flag is refused by every entry point (the ladder is constant-time as it is);
                                      double_scalar_mult / ecdsa_verify* take no flags: they are for public data */
  ECSIMD_HIP_LADDER_RADIX32 = 256, /* ladder only (r4): run the 254 iterations on 8 x 32-bit canonical words (the kernel of rounds 1-3) instead of the
                                      reduced-radix loop (nine signed 29-bit limbs, carry-free columns: fe29.cuh) that is the default since round 4.
                                      The field values of every iteration are the same, so X, Y, Z are bit-identical; kept for A/B measurements.
                                      REF_SQUARE_COMPAT implies it (the dropped carry depends on the 32-bit Montgomery digits).  Same constant-time shape */
  ECSIMD_HIP_ALG_WINDOWED_BIG = 32 /* scalar_mult_base + OUT_AFFINE: 20-bit windows with odd digits over a 436 MB table of the odd
                                      multiples (2d+1)*2^(20i)*G (13 windows x 2^19 entries) in device memory, built on first
                                      use (0.23 s per curve): 12 mixed additions per scalar; same results.  NOT for secret scalars:
                                      each window is a 64-byte read from device memory at an address formed from 20 scalar bits */
};

/* ---- context, stream and memory ------------------------------------------------------- */
int ecsimd_hip_init(int device, ecsimd_hip_ctx** ctx);
int ecsimd_hip_destroy(ecsimd_hip_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream).  NULL is HIP's default (null)
 * stream -- which is what torch's default stream is -- not "none".  The context's scratch memory is ordered
 * by its stream: switching makes the new stream wait (by event, no host synchronisation) for the work this
 * context enqueued on the previous one, which must still exist at that moment.  Streams under hipGraph capture are
 * selected without that hand-off; every compute entry point is capturable once its first call has sized the
 * context workspace and built its tables (those steps allocate and synchronise).  A call that would have to grow the
 * workspace or build a table WHILE its stream is being captured returns ECSIMD_HIP_ERR_BAD_ARG instead (growing frees the
 * old block, which an earlier capture may still point into): warm up at the largest batch size, then capture. */
int ecsimd_hip_set_stream(ecsimd_hip_ctx* ctx, void* hip_stream);
/* Go back to the non-blocking stream the context created in ecsimd_hip_init (the default). */
int ecsimd_hip_use_own_stream(ecsimd_hip_ctx* ctx);
int ecsimd_hip_sync(ecsimd_hip_ctx* ctx);
/* Reference-square compatibility for every entry point that evaluates the reference's own expression DAG and contains a
 * squaring (square, mgry_sqr, mgry_pow, gfp_inverse, gfp_sqrt, compute_y, to_affine, DBLU ... TRPLU, the ladder):
 * on != 0 makes them square with mul.h:160-212 as written, dropped carry included (see ECSIMD_HIP_REF_SQUARE_COMPAT), and
 * walk exponents bit by bit as mgry_ops.h:44-86 does, so a caller gets the compiled reference's bits on every input.
 * Off by default.  The windowed algorithms are not the reference's and refuse to run while it is on. */
int ecsimd_hip_set_ref_square_compat(ecsimd_hip_ctx* ctx, int on);
/* The option as it stands: 0 / 1 (a caller that runs one job on two contexts mirrors it: integration/scalar_mult_p256_adapter.cpp); < 0: ctx is NULL. */
int ecsimd_hip_get_ref_square_compat(const ecsimd_hip_ctx* ctx);
const char* ecsimd_hip_last_error(const ecsimd_hip_ctx* ctx);
const char* ecsimd_hip_version(void);
int ecsimd_hip_malloc(ecsimd_hip_ctx* ctx, void** dptr, size_t bytes);
int ecsimd_hip_free(ecsimd_hip_ctx* ctx, void* dptr);
int ecsimd_hip_memcpy_h2d(ecsimd_hip_ctx* ctx, void* dst, const void* src, size_t bytes);
int ecsimd_hip_memcpy_d2h(ecsimd_hip_ctx* ctx, void* dst, const void* src, size_t bytes);
/* device -> device on the context's stream (asynchronous, ordered with the kernels); what a by-value copy of a
 * reference `wide` (bignum.h:38-102: registers, copied freely) costs here when an in-out parameter is updated. */
int ecsimd_hip_memcpy_d2d(ecsimd_hip_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Curve constants as the engine uses them (host memory, 4 x u64 each):
 * which = 0 p, 1 a, 2 b, 3 Gx, 4 Gy, 5 R mod p, 6 R^2 mod p, 7 -R mod p, 8 a*R, 9 b*R, 10 p-2, 11 (p+1)/4
 * (mgry_csts.h:15-24, curve_group.h:31-32, gfp.h:79-87). */
int ecsimd_hip_get_constant(int curve, int which, uint64_t out[4]);
/* A field id for the odd modulus p >= 3 (host pointer, 4 x u64 little-endian limbs): the host derives R mod p, R^2 mod p, -R mod p and m' = -p^-1 mod 2^32
 * (mgry_csts.h:15-35, mgry_mul.h:33-38) once; the element-wise field entry points then accept the id as `curve` (for a field id get_constant
 * reads zero in the curve slots 1-4, 8, 9).  The same (p, flags) gives the same id (the two curve primes give 0 / 1 whatever the flags); the same p with and
 * without MODULUS_PRIME are two ids, so that nobody's gfp_inverse changes under them; ids live as long as the process; thread-safe.  flags: 0 or ECSIMD_HIP_MODULUS_PRIME.  gfp_sqrt needs p = 3 mod 4, as the reference's GFp does (gfp.h:84). */
int ecsimd_hip_register_modulus(const uint64_t p[4], int flags, int* field_id);

/* A curve id for y^2 = x^3 + a x + b over GF(p) with generator (gx, gy) -- the reference's curve_group<Curve> instantiated with any Curve type that has
 * bn_type, P, A, B, Gx, Gy (curve.h:12-15; curve_group.h:25-33 derives Am, Bm from the type, :64-87 DBLU takes a from it, :91-218 the co-Z formulas and the
 * ladder are curve-independent) -- as a run-time registration: host pointers, 4 x u64 little-endian limbs, classical values < p.  p must be a prime with
 * p = 3 mod 4 (what the reference's GFp<WBN, P> needs: gfp.h:84) and the caller vouches for its primality; the generator must lie on the curve and the
 * curve must be non-singular (checked: ECSIMD_HIP_ERR_BAD_ARG).  n = the group order, or NULL (the reference has no order either: its ladder takes any
 * 256-bit k; with it the id gains the entry points listed below).  The id (>= ECSIMD_HIP_FIRST_REGISTERED_CURVE; the same parameters -- the order, or its
 * absence, included -- give the same id, so nobody's registration changes what an id somebody else holds does; P-256's or secp256k1's
 * parameters give 0 / 1 unless flags = ECSIMD_HIP_CURVE_GENERIC_KERNELS, which registers them like any other curve -- how tests hold the generic kernels
 * to the special-form ones bit for bit) is accepted by
 *     from_affine, to_affine, compute_y, on_curve, dblu, zaddu, zdau, add_z2_1, trplu, zdau_repeat, scalar_mult, scalar_mult_1s, scalar_mult_base, scalar_mult_host
 * (flags BASE_* | OUT_* | LADDER_RADIX32 | REF_SQUARE_COMPAT: the reference's ladder; of the table-driven algorithms a registered curve has all but the GLV split, all for
 * a curve registered with its order n >= 2^255 and all with OUT_AFFINE only: scalar_mult_base with ALG_WINDOWED -- a 4-bit odd-digit table of multiples of
 * ITS generator in LDS, built from the ladder on first use -- with ALG_WINDOWED | ALG_CONSTANT_TIME -- 5-bit windows, 52 x 16 odd multiples in 53 KB of LDS,
 * every entry of a window read and one kept under lane masks: SAFE for secret scalars -- or with ALG_WINDOWED_SIGNED -- signed 7-bit windows, 37 x 64 odd
 * multiples in 148 KiB of LDS, 36 additions instead of 63, public scalars (1.6 x the 4-bit comb) -- or with ALG_WINDOWED_BIG -- 20-bit windows, 13 x 2^19 odd
 * multiples in 436 MB of DEVICE memory per curve and context (built on first use: 0.3 s, 1.3 GB of temporary memory), 12 additions, public scalars (3.7 x) --
 * and scalar_mult / scalar_mult_1s with ALG_WINDOWED on a variable base --
 * the lane's own table of the eight odd multiples of P over one Z, 63 windows of three doublings and a fused double-add in modified Jacobian coordinates
 * on the isomorphic curve (any coefficient a; 640 B of context workspace per element, 2^22 at a time; 1.4 x the ladder's rate) -- public scalars: the table
 * is indexed by the scalar's digits; with ALG_CONSTANT_TIME every entry of the lane's table is read in every window and one kept under lane masks: SAFE for
 * secret scalars like the built-in curves' form (1.3 x the ladder).  The group must have prime order (cofactor 1: every valid point then has order n, which
 * is what "no addition inside the loop is exceptional" rests on) -- CHECKED at registration (n in p's Hasse interval, Miller-Rabin: ecsimd_hip_curve_capabilities);
 * an id without it keeps the ladder for a variable base (BAD_ARG for the flag, ladder passes inside double_scalar_mult / ecdsa_verify).
 * Either returns the true k P for every k, (0, 0) for k = 0 mod n.  The other ALG_* shapes exist for the two built-in curves only), by affine_add, sec1_encode, sec1_decode and -- when n was given, p < 2n, and
 * n - u is a good ladder scalar for u in {n - 1, 2^256 - n - 1, 2^256 - n} (every prime-order curve of this size) -- by double_scalar_mult,
 * ecdsa_verify_rx, ecdsa_verify, ecdsa_sign: u1 G comes from the generator's signed comb -- from the 20-bit comb once that exists or the batch reaches 2^20 -- (sign: k G from the constant-time 5-bit comb), u2 Q (public) from the
 * lane's window table -- correct for every scalar in [0, n); ecdsa_sign's scratch is zeroed like the built-in curves'.  (n < 2^255: passes of the ladder
 * instead, the scalars kept clear of its three degenerate values: u -> n - u and the result negated.)  Like the built-in curves', a call of
 * scalar_mult_base(OUT_AFFINE) without an algorithm flag on up to 2^16 lanes takes that comb (constant time) and returns the ladder's affine bits, its
 * three degenerate scalars included.  And, as a FIELD id, by every element-wise field entry point.  Same level-J parity as the built-in curves: X, Y, Z are
 * the bits the reference instantiated with this Curve returns.  The ladder's 254 iterations run on nine signed 29-bit limbs with the dense p in SGPRs
 * (81 multiply-adds per reduction where P-256's sparse form has 36); LADDER_RADIX32 / REF_SQUARE_COMPAT run them on 8 x 32-bit canonical words.
 * Process-wide, thread-safe, ids live as long as the process. */
enum { ECSIMD_HIP_CURVE_GENERIC_KERNELS = 1 };
/* What an id can do beyond the reference's layers (host only, no context): caps = a mask of
 *   HAS_ORDER (registered with n), COMB (ALG_WINDOWED [| ALG_CONSTANT_TIME] on its generator: n >= 2^255), ECDSA (double_scalar_mult, ecdsa_*: p < 2n and the
 *   ladder's degenerate scalars have good images), WINDOW_VARIABLE_BASE (ALG_WINDOWED [| ALG_CONSTANT_TIME] on a variable base: n >= 2^255, n lies in p's Hasse
 *   interval -- so the group HAS order n -- and n passes Miller-Rabin: every point but infinity has order n).  The two built-in ids report all four. */
enum { ECSIMD_HIP_CURVE_HAS_ORDER = 1, ECSIMD_HIP_CURVE_COMB = 2, ECSIMD_HIP_CURVE_ECDSA = 4, ECSIMD_HIP_CURVE_WINDOW_VARIABLE_BASE = 8 };
int ecsimd_hip_curve_capabilities(int curve, int* caps);
int ecsimd_hip_register_curve(const uint64_t p[4], const uint64_t a[4], const uint64_t b[4], const uint64_t gx[4], const uint64_t gy[4],
                              const uint64_t n[4], int flags, int* curve_id);

/* ---- L2: bignum ops (curve independent) ----------------------------------------------- */
/* add.h:11-34  add: out = a + b mod 2^256, carry[i] = carry-out (carry may be NULL) */
int ecsimd_hip_add(ecsimd_hip_ctx*, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* carry, size_t n);
/* sub.h:12-38  sub: out = a - b mod 2^256, borrow[i] = borrow-out */
int ecsimd_hip_sub(ecsimd_hip_ctx*, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* borrow, size_t n);
/* sub.h:46-75  sub_if_above: out = a >= p ? a - p : a (p is per element, like the reference's wide p) */
int ecsimd_hip_sub_if_above(ecsimd_hip_ctx*, const uint64_t* a, const uint64_t* p, uint64_t* out, size_t n);
/* Lane masks on the device (the reference's eve::logical<wide>, bignum.h:136-137, with eve::all / eve::any):
 * cmp_eq: flag[i] = (a[i] == b[i]) over the low `limbs` (1..8) limbs of each element (element stride 4 limbs,
 * 8 for limbs > 4: a 128-bit value occupies the low half of a 256-bit element); mask_op: NOT (b ignored), AND, OR, EQ of
 * 0/1 byte masks; mask_count: number of non-zero flags (synchronises the stream: all = count == n, any = count > 0). */
enum { ECSIMD_HIP_MASK_NOT = 0, ECSIMD_HIP_MASK_AND = 1, ECSIMD_HIP_MASK_OR = 2, ECSIMD_HIP_MASK_EQ = 3 };
int ecsimd_hip_cmp_eq(ecsimd_hip_ctx*, const uint64_t* a, const uint64_t* b, int limbs, uint8_t* flag, size_t n);
int ecsimd_hip_mask_op(ecsimd_hip_ctx*, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);
int ecsimd_hip_mask_count(ecsimd_hip_ctx*, const uint8_t* a, size_t n, size_t* count);
/* cmp.h:11-13  cmp_lt: flag[i] = a < b */
int ecsimd_hip_cmp_lt(ecsimd_hip_ctx*, const uint64_t* a, const uint64_t* b, uint8_t* flag, size_t n);
/* shift.h:13-32  shift_left_one: out = a << 1 mod 2^256, carry[i] = bit 255 of a */
int ecsimd_hip_shift_left_one(ecsimd_hip_ctx*, const uint64_t* a, uint64_t* out, uint8_t* carry, size_t n);
/* mul.h:150-158  mul: out8 = a * b (512 bits) */
int ecsimd_hip_mul(ecsimd_hip_ctx*, const uint64_t* a, const uint64_t* b, uint64_t* out8, size_t n);
/* mul.h:214-221  square: out8 = a * a */
int ecsimd_hip_square(ecsimd_hip_ctx*, const uint64_t* a, uint64_t* out8, size_t n);
/* swap.h:15-22  swap_if: swap a[i] and b[i] in place where mask[i] != 0 */
int ecsimd_hip_swap_if(ecsimd_hip_ctx*, const uint8_t* mask, uint64_t* a, uint64_t* b, size_t n);
/* ifelse.h:15-22  if_else: out[i] = mask[i] != 0 ? a[i] : b[i] (out may alias a or b) */
int ecsimd_hip_if_else(ecsimd_hip_ctx*, const uint8_t* mask, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);

/* ---- wire formats on the device ------------------------------- */
/* serialization.h:12-24 bn_from_bytes_BE / :26-48 bn_to_bytes_BE over a batch: 32 big-endian bytes per
 * element <-> 4 x u64 little-endian limbs.  `bytes` is a device pointer, 16-byte aligned. */
int ecsimd_hip_from_bytes_be(ecsimd_hip_ctx*, const uint8_t* bytes, uint64_t* out, size_t n);
int ecsimd_hip_to_bytes_be(ecsimd_hip_ctx*, const uint64_t* in, uint8_t* bytes, size_t n);
/* The reference's own data format on the device (r5).  Its wide_bignum<bignum_256> is eve::wide<bignum, fixed<4>> -- four 256-bit lanes stored limb-major,
 * u64[limb * 4 + lane], 128 bytes (bignum.h:99-100; eve/arch/cpu/as_register.hpp:55-60) -- and a wide_jacobian_curve_point is three of them
 * (jacobian_curve_point.h:14-58).  A caller that keeps reference types copies its array of wides / points to the device as it is; these two do the 4 x 4
 * transposition there, at HBM speed, instead of a per-lane loop on the host: wide w of a record array (`record_bytes` apart, the wide at `offset_bytes`
 * inside its record: 128 / 0 for an array of wides; 384 / 0, 128, 256 for x, y, z of an array of Jacobian points) <-> elements 4w .. 4w + 3 of an ABI
 * array (u64[4 * e + limb]).  `wides` is a device pointer, 8-byte aligned; record_bytes and offset_bytes multiples of 8; not in place.
 * integration/scalar_mult_p256_adapter.cpp is the caller. */
int ecsimd_hip_wide4_to_lanes(ecsimd_hip_ctx*, const void* wides, size_t record_bytes, size_t offset_bytes, uint64_t* out, size_t n_wides);
int ecsimd_hip_lanes_to_wide4(ecsimd_hip_ctx*, const uint64_t* in, void* wides, size_t record_bytes, size_t offset_bytes, size_t n_wides);
/* utility.h:45-51 wide_mask_bit: flag[i] = bit `bit` (0..255, 0 = least significant) of a[i] */
int ecsimd_hip_mask_bit(ecsimd_hip_ctx*, const uint64_t* a, int bit, uint8_t* flag, size_t n);
/* SEC 1 v2 2.3.3: affine classical (x, y) -> 04||X||Y (65 B per point) or, compressed, 02/03||X (33 B). */
int ecsimd_hip_sec1_encode(ecsimd_hip_ctx*, int curve, const uint64_t* x, const uint64_t* y, uint8_t* out, size_t n, int compressed);
/* SEC 1 v2 2.3.4: the inverse.  ok[i] = 1 iff the record is well formed, x (and y) < p and the point is on
 * the curve (uncompressed) / x^3 + a x + b is a square (compressed: y recovered with the prefix's parity;
 * generalises curve_point_ops.h:12-22 from_x to per-lane validity and both curves).  The point at infinity
 * (single byte 00) has no fixed-size record and is not representable here. */
int ecsimd_hip_sec1_decode(ecsimd_hip_ctx*, int curve, const uint8_t* in, uint64_t* x, uint64_t* y, uint8_t* ok, size_t n, int compressed);

/* ---- L3: GF(p) ------------------------------------------------------------------------
 * `curve` is a FIELD id in this section (enum ecsimd_hip_field): a curve's prime or any registered modulus. */
/* modular.h:10-15 mod_add, :24-41 mod_sub, mgry_ops.h:14-22 mgry_shift_left<count> (1 <= count <= 255).
 * count | ECSIMD_HIP_SHIFT_FUSED runs pairs of doublings as one quadrupling with a single conditional subtraction (what the point
 * formulas use on their own intermediates): the same canonical residue for every a < p; for a >= p only the plain form is
 * the reference's chain of doublings. */
enum { ECSIMD_HIP_SHIFT_FUSED = 0x100 };
int ecsimd_hip_mod_add(ecsimd_hip_ctx*, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int ecsimd_hip_mod_sub(ecsimd_hip_ctx*, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int ecsimd_hip_mod_shift_left(ecsimd_hip_ctx*, int curve, const uint64_t* a, int count, uint64_t* out, size_t n);
/* Extension (the reference stops at mod_add / mod_sub, modular.h): classical a * b mod p, canonical. */
int ecsimd_hip_mod_mul(ecsimd_hip_ctx*, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* mgry_mul.h:84-121 details::mgry_reduce<P>: out = a8 * 2^-256 mod p (a8 < p * 2^256) */
int ecsimd_hip_mgry_reduce(ecsimd_hip_ctx*, int curve, const uint64_t* a8, uint64_t* out, size_t n);
/* mgry_ops.h:31-42 mgry_mul / mgry_sqr */
int ecsimd_hip_mgry_mul(ecsimd_hip_ctx*, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int ecsimd_hip_mgry_sqr(ecsimd_hip_ctx*, int curve, const uint64_t* a, uint64_t* out, size_t n);
/* mgry.h:47-55 from_classical / to_classical */
int ecsimd_hip_mgry_from_classical(ecsimd_hip_ctx*, int curve, const uint64_t* a, uint64_t* out, size_t n);
int ecsimd_hip_mgry_to_classical(ecsimd_hip_ctx*, int curve, const uint64_t* a, uint64_t* out, size_t n);
/* mgry_ops.h:44-86 mgry_pow: out = a^e (Montgomery), e = ONE public exponent (host pointer, 4 x u64) */
int ecsimd_hip_mgry_pow(ecsimd_hip_ctx*, int curve, const uint64_t* a, const uint64_t exponent[4], uint64_t* out, size_t n);
/* gfp.h:42-44 inverse, :60-64 opposite, :46-54 sqrt.  sqrt reports validity PER ELEMENT in ok[]
 * (the reference collapses a wide to all-or-nothing; the C++ header reproduces that on top).  inverse shares one
 * inversion among up to 128 elements (Montgomery's trick) when out and a are different buffers; out == a inverts every
 * element on its own; 0 maps to 0 either way.  The inversion itself is the constant-time division-step ("safegcd") one,
 * not a^(p-2): the same unique inverse for every CANONICAL operand (a < p), which is what these two entry points take. */
int ecsimd_hip_gfp_inverse(ecsimd_hip_ctx*, int curve, const uint64_t* a, uint64_t* out, size_t n);
int ecsimd_hip_gfp_opposite(ecsimd_hip_ctx*, int curve, const uint64_t* a, uint64_t* out, size_t n);
int ecsimd_hip_gfp_sqrt(ecsimd_hip_ctx*, int curve, const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n);

/* ---- L4/L5: points and the group ------------------------------------------------------ */
/* jacobian_curve_point.h:25-31 from_affine (Z := R mod p), :33-42 to_affine.  to_affine uses Montgomery's
 * simultaneous inversion (one shared inversion per up to 128 elements (k_affine.inc BATCH_INVERSION_MAX); identical values) unless x/y alias the inputs.
 * to_affine: y may be NULL (the x coordinate only: 5 field multiplications per element instead of 7). */
int ecsimd_hip_from_affine(ecsimd_hip_ctx*, int curve, const uint64_t* x, const uint64_t* y, uint64_t* jx, uint64_t* jy, uint64_t* jz, size_t n);
int ecsimd_hip_to_affine(ecsimd_hip_ctx*, int curve, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n);
/* curve_group.h:43-58 compute_y for y^2 = x^3 + a x + b (classical in/out), per-element ok[] */
int ecsimd_hip_compute_y(ecsimd_hip_ctx*, int curve, const uint64_t* x, uint64_t* y, uint8_t* ok, size_t n);
/* curve_group.h:64-87 DBLU: r = 2P, P rewritten in place with r's Z.  P.z must be mgry(1).
 * (DBLU, ZADDU, ZDAU, TRPLU: co-Z, ONE Z for the result and the rewritten operand -- rz may be the very array pz (ZDAU: qz); it is then written once.)
 * Point formulas and the ladder take CANONICAL coordinates (< p), which is what every entry point of this library returns; the element-wise
 * field operations above additionally reproduce the reference on operands >= p (it never rejects them, tests/ops.cpp:232). */
int ecsimd_hip_dblu(ecsimd_hip_ctx*, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
/* curve_group.h:91-116 ZADDU: r = P + O (co-Z), P rewritten in place with r's Z. */
int ecsimd_hip_zaddu(ecsimd_hip_ctx*, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, const uint64_t* ox, const uint64_t* oy, const uint64_t* oz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
/* curve_group.h:120-153 ZDAU: r = 2P + Q (co-Z), Q rewritten in place with r's Z. */
int ecsimd_hip_zdau(ecsimd_hip_ctx*, int curve, const uint64_t* px, const uint64_t* py, const uint64_t* pz, uint64_t* qx, uint64_t* qy, uint64_t* qz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
/* Diagnostic / micro-benchmark (r4): ZDAU applied `iters` times IN REGISTERS -- the ladder's loop body alone (curve_group.h:206-210 without the
 * scalar): per iteration (P, Q) <- (2P + Q, Q re-expressed with the new Z), and after iteration t the two outputs are exchanged where bit
 * (t mod 64) of `swap_bits` is set.  (rx, ry) = the final P, (sx, sy) = the final Q, oz their Z; all Montgomery form, canonical.
 * radix = 32: on field.cuh's 8 x 32-bit canonical words (rounds 1-3); radix = 29: on fe29.cuh's nine signed 29-bit limbs, the representation
 * the ladder's loop runs in since round 4 -- the SAME field values, so the same bits out (tests/test_gpu_parity.py pins both to the oracle's
 * ZDAU iterated on the CPU).  Not available with the reference-square option for radix 29. */
int ecsimd_hip_zdau_repeat(ecsimd_hip_ctx*, int curve, const uint64_t* px, const uint64_t* py, const uint64_t* pz, const uint64_t* qx, const uint64_t* qy,
                           uint64_t* rx, uint64_t* ry, uint64_t* sx, uint64_t* sy, uint64_t* oz, size_t n, int iters, uint64_t swap_bits, int radix);
/* curve_group.h:155-179 ADD_Z2_1: r = A + B with B = (bx, by) Montgomery-form affine (Z2 = mgry(1)). */
int ecsimd_hip_add_z2_1(ecsimd_hip_ctx*, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
/* Extension: the complete form of the mixed addition -- r = A + B for every input: A = infinity (Z = 0), B = infinity
 * ((0, 0)), A = B (tangent), A = -B (r = infinity: Z = 0).  Same conventions as add_z2_1 (B Montgomery-form affine);
 * r is a representative of the same point as ADD_Z2_1's where that formula is defined (Z3 = Z1*H instead of 2*Z1*H). */
int ecsimd_hip_add_mixed_complete(ecsimd_hip_ctx*, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);
/* curve_group.h:183-186 TRPLU: r = 3P, P rewritten in place with r's Z.  P.z must be mgry(1). */
int ecsimd_hip_trplu(ecsimd_hip_ctx*, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n);

/* curve_group.h:189-218 scalar_mult (per-element scalar k[i], per-element base point (x[i], y[i])):
 * the co-Z Joye double-add ladder of the reference, any 256-bit k.  flags = BASE_* | OUT_*, optionally
 * OUT_AFFINE | ALG_WINDOWED for the faster windowed algorithm (affine-level parity, see the flag). */
int ecsimd_hip_scalar_mult(ecsimd_hip_ctx*, int curve, const uint64_t* k, const uint64_t* x, const uint64_t* y,
                           uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags);
/* curve_group.h:221-251 scalar_mult_1s: ONE scalar (host pointer) for all points; flags as for scalar_mult. */
int ecsimd_hip_scalar_mult_1s(ecsimd_hip_ctx*, int curve, const uint64_t k1[4], const uint64_t* x, const uint64_t* y,
                              uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags);
/* curve_group.h:35-41 WJG + scalar_mult: k[i] * G (base = the curve generator); ecsimd_hip_scalar_mult with x = y = NULL and ECSIMD_HIP_BASE_GENERATOR is the same call.
 * Small batches (r4): with OUT_AFFINE, no ALG_* / REF_SQUARE_COMPAT / LADDER_RADIX32 flag and n <= 2^16 the product comes from the constant-time
 * 5-bit comb (ALG_WINDOWED | ALG_CONSTANT_TIME's kernel: as safe for secret scalars as the ladder) instead of a 254-iteration ladder launch whose
 * cost does not depend on n -- 0.2 ms instead of 1.3 ms -- with the SAME affine bits: at the ladder's three degenerate scalars (n - 1,
 * 2^256 - n - 1, 2^256 - n) the lanes take the ladder's own coordinates by select.  The first such call per curve builds the comb's table. */
int ecsimd_hip_scalar_mult_base(ecsimd_hip_ctx*, int curve, const uint64_t* k, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags);
/* Extensions built on the kernels above (not in the reference):
 * affine_add: R = A + B for affine classical points, one shared inversion per up to 128 points (Montgomery's trick).  (0, 0) encodes the point
 * at infinity on input and output; finite[i] = 0 marks an infinite result (finite and ry may be NULL; rx must
 * not alias an input). */
int ecsimd_hip_affine_add(ecsimd_hip_ctx*, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* bx, const uint64_t* by,
                          uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n);
/* on_curve: ok[i] = x[i], y[i] < p and y^2 = x^3 + a x + b (classical coordinates) -- SEC 1 public-key validation (both
 * curves have cofactor 1); (0, 0), this library's point at infinity, fails. */
int ecsimd_hip_on_curve(ecsimd_hip_ctx*, int curve, const uint64_t* x, const uint64_t* y, uint8_t* ok, size_t n);
/* double_scalar_mult: R[i] = u1[i]*G + u2[i]*Q[i], affine classical (the ECDSA-verification shape; pass
 * ry = NULL for x only).  u1*G uses the windowed fixed-base kernel, u2*Q the windowed variable-base kernels
 * (ALG_WINDOWED): correct for every 256-bit u1, u2; 1 632 B of context workspace per element, 2^22 at a time.
 * Q[i] is VALIDATED (on_curve): an invalid public key -- off the curve, a coordinate >= p, the point at infinity --
 * gives R[i] = (0, 0) and finite[i] = 0.  The scalars here are public (signature verification): the window kernels index
 * their tables by scalar digits and are not for secret scalars.  u1*G goes through the 436 MB table of 20-bit windows once it
 * exists in this context or n >= 2^16 (which builds it); smaller batches use the 148 KiB table in LDS and keep nothing that size. */
int ecsimd_hip_double_scalar_mult(ecsimd_hip_ctx*, int curve, const uint64_t* u1, const uint64_t* u2, const uint64_t* qx, const uint64_t* qy,
                                  uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n);
/* (ok[i] is also 0 when Q[i] fails the validation above.)  The acceptance test of an ECDSA verification for precomputed u1 = e/s, u2 = r/s (mod n; ecsimd_hip_ecdsa_verify
 * below computes them on the device): ok[i] = 1 iff u1[i]*G + u2[i]*Q[i] is a finite point whose
 * x coordinate, reduced mod n, equals r[i].  Same workspace as double_scalar_mult plus 33 B per element. */
int ecsimd_hip_ecdsa_verify_rx(ecsimd_hip_ctx*, int curve, const uint64_t* u1, const uint64_t* u2, const uint64_t* qx, const uint64_t* qy,
                               const uint64_t* r, uint8_t* ok, size_t n);
/* ECDSA verification of n signatures (SEC 1 v2 4.1.4, FIPS 186-5 6.4.2; not in the reference -- SURVEY.md 8(f) rank 4, "the first real application of the
 * engine").  Per element, all classical 4 x u64 LE limbs in device memory: e = the message digest as an integer (its leftmost 256 bits; ANY 256-bit value is
 * taken and reduced mod n), (r, s) = the signature, (qx, qy) = the public key.  ok[i] = 1 iff 1 <= r, s < n, Q is a valid public key (on_curve), and
 * R = (e/s) G + (r/s) Q is a finite point with x(R) mod n == r.  The arithmetic modulo the group order n runs on the device (field id
 * ECSIMD_HIP_FIELD_*_ORDER: generic Montgomery multiplication, one constant-time division-step inversion shared by up to 128 signatures), then
 * ecdsa_verify_rx's kernels.  Public data only (the window kernels index tables by scalar digits).  Workspace: ecdsa_verify_rx's plus 65 B per element. */
int ecsimd_hip_ecdsa_verify(ecsimd_hip_ctx*, int curve, const uint64_t* e, const uint64_t* r, const uint64_t* s, const uint64_t* qx, const uint64_t* qy,
                            uint8_t* ok, size_t n);
/* ECDSA signing of n digests (SEC 1 v2 4.1.3; not in the reference): e = the digest as an integer, d = the private key, k = the per-signature nonce --
 * the CALLER's (RFC 6979 or a DRBG: this library has no hash and no random source), 1 <= d, k < n.  R = k G comes from the constant-time comb (the kernel
 * of ALG_WINDOWED | ALG_CONSTANT_TIME), r = x(R) mod n, s = k^-1 (e + r d) mod n on the device (generic Montgomery products on the order's field id, one
 * constant-time division-step inversion shared by up to 128 signatures).  ok[i] = 1 and (r, s) a valid signature, or ok[i] = 0 and r = s = 0 where d or k is
 * out of range or r or s came out 0 (probability ~2^-255: sign again with another nonce).  d and k are treated as SECRETS: no branch or address depends on
 * them, checked on the shipped ISA: the comb's window loop by tools/ct_check.py check(), and the comb as a whole, the x-only simultaneous inversion of k G and
 * the scalar-field kernel (its division-step inversion included) by check_secret_flow, a taint analysis from the loads of d, k and k G to every branch
 * condition, lane mask in force at a memory access, and address (tests/test_constant_time_isa.py).  The Jacobian k G and its x are zeroed in the context
 * workspace on the stream before the call returns.  r, s must not alias an input.  Workspace: 128 B per element. */
int ecsimd_hip_ecdsa_sign(ecsimd_hip_ctx*, int curve, const uint64_t* e, const uint64_t* d, const uint64_t* k, uint64_t* r, uint64_t* s, uint8_t* ok, size_t n);
/* Diagnostic (r5): ONE function of the reduced-radix layer the multiplication-bound loops run on (fe29.cuh: nine signed 29-bit limbs in 32-bit words,
 * Montgomery radix 2^261, lazy carries) on RAW operands -- int32 limbs exactly as a loop holds them between two operations: element e's coordinate c, limb l at
 * in[(e * NIN + c) * 9 + l] (device memory).  What tests use to hand the device the states and operand pairs at which the interval proofs of
 * tools/radix29_model.py reach their largest 64-bit columns (tests/golden/fe29_witnesses.json) and compare the result limb for limb with the exact model; no
 * other entry point can produce such inputs (their operands enter the loops as tight limbs).  op (NIN -> NOUT coordinates):
 *   0 zdau29 (x1, x2, dx, y1, dy, z -> the same six; swap != 0 exchanges the two output points)      1 madd29 (X, Y, Z, x2, y2 -> X, Y, Z)      2 jdbl29 (X, Y, Z ->)
 *   3 dbl_add29 (X, Y, Z, x2, y2 ->)    4 madd29_hr + madd29v_finish    5 pdbl29, 6 padd29 (secp256k1: the complete law)    7 mul29 (a, b -> r)    8 sqr29 (a -> r)
 *   9 gjdbl29 (X, Y, Z, W = a Z^4 -> the same four; swap != 0: the doubled point's W is formed)    10 zaddu29 (x1, y1, x2, y2, z -> rx, ry, x1', y1', z', dx)
 * curve: 0 / 1 (for secp256k1 the loops' own domain: values x * 2^261 of the CLASSICAL x; ops 0-8, secp256k1 also 10), or a registered curve id (ops 0, 1, 3, 7, 8, 9,
 * 10: the dense reduction; the window loop of k_gvarwin.hip).
 * Operands outside the proven bounds give whatever 32- / 64-bit wrap-around gives. */
int ecsimd_hip_fe29_raw(ecsimd_hip_ctx* ctx, int curve, int op, const int32_t* in, int32_t* out, size_t n, int swap);
/* Diagnostic: the context's grow-only scratch block (device pointer and size; NULL / 0 before the first call that needed one).  What a test reads
 * back to see that ecdsa_sign left no nonce-derived data behind (tests/test_gpu_fields.py); valid until the next call that grows the block. */
int ecsimd_hip_workspace_info(ecsimd_hip_ctx* ctx, const void** dptr, size_t* bytes);
/* scalar_mult with every array in HOST memory (n elements each; pageable is fine): the PCIe-inclusive form of the hot path.  Same curve ids, flags and
 * meaning as ecsimd_hip_scalar_mult (x = y = NULL with ECSIMD_HIP_BASE_GENERATOR: k G; OUT_AFFINE: oz unused, oy optional).  Chunks of 2^19 elements
 * alternate between the context and a helper context it creates on first use (a second stream): the copies of one chunk overlap the ladder of its
 * neighbour.  Synchronous -- returns with the results in place; not capturable.  One MI355X, 2^22 elements of pageable memory: 56 M scalar mults/s against
 * 58 with the arrays resident in HBM (reuse the output arrays: fresh pages are first touched inside the call).  The reference's own types instead of arrays: integration/scalar_mult_p256_adapter.cpp. */
int ecsimd_hip_scalar_mult_host(ecsimd_hip_ctx*, int curve, const uint64_t* k, const uint64_t* x, const uint64_t* y,
                                uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags);
/* lib/scalar_mult_p256.cpp:10-12: scalar_mult_p256(x, P) -- P-256, base in Montgomery form with
 * Z = mgry(1), Jacobian Montgomery output. */
int ecsimd_hip_scalar_mult_p256(ecsimd_hip_ctx*, const uint64_t* k, const uint64_t* xm, const uint64_t* ym,
                                uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n);

/* ---- synthetic inputs and measurement (bench / tests) --------------------------------- */
/* word w of element i of stream s = splitmix64(seed ^ (s << 56) ^ (4 i + w));
 * clear_top_bits > 0 clears that many top bits of limb 3 (field elements < 2^255 < p). */
int ecsimd_hip_fill_random(ecsimd_hip_ctx*, uint64_t* out, size_t n, uint64_t seed, uint64_t stream, uint64_t first_index, int clear_top_bits);
/* Dependency-free v_mad_u64_u32 stream: the integer-multiply roofline denominator.  Returns the
 * number of mad32 executed in *mads; elapsed device time in *ms (HIP events on the ctx stream). */
int ecsimd_hip_peak_mad32(ecsimd_hip_ctx*, int iters, double* mads, double* ms);

/* ---- device groups: one batch over several GPUs -------------------------------
 * The reference is single-threaded and has no collectives (nothing to cite); BASELINE.json north_star asks for
 * batches sharded across the GPUs of a node "with RCCL over xGMI only for the final gather", host code behind this ABI.
 * A group owns one context per listed device.  Member m of G owns the contiguous slice shard_range(n, m, G) of a
 * batch (the first n % G members take one element more); no collective touches the data path; the result shards
 * are gathered to member 0's device memory by one grouped ncclSend / ncclRecv exchange on a second stream of member 0's
 * device, so that member 0's next ladder overlaps it.  RCCL is looked up at run time when the group has more than one
 * device -- the copy the process already has (soname librccl.so.1, e.g. torch's) if there is one, else ROCm's; the library
 * needs RCCL neither to build nor to link.  A device may be listed twice, in which case those members exchange
 * by device copies (how a one-GPU machine exercises the bookkeeping).  One host thread drives a group; every group
 * entry point leaves the caller's current HIP device as it found it (the per-context entry points above select their
 * context's device and leave it selected).
 *
 * EXPERIMENTAL until a machine with two or more GPUs has run tests/test_gpu_parity.py::test_device_group_behind_the_c_abi
 * with distinct devices: the bookkeeping, staging, offsets and the device-copy gather run in every GPU test run, RCCL is
 * loaded and exercised on a one-rank communicator (ecsimd_hip_group_rccl_selftest), but the ncclCommInitAll +
 * multi-device ncclSend / ncclRecv exchange itself has only been checked against RCCL's documented contract. */
typedef struct ecsimd_hip_group ecsimd_hip_group;
enum {
  ECSIMD_HIP_GROUP_NO_GATHER = 0x10000   /* ecsimd_hip_group_scalar_mult only: compute, leave every shard where its member wrote it
                                            (member 0: its slice of ox/oy/oz; the others: group staging).  For timing the ladders
                                            without the exchange ("throughput with and without the gather") */
};
int ecsimd_hip_shard_range(size_t n_total, int member, int members, size_t* first, size_t* count);   /* pure host arithmetic */
int ecsimd_hip_group_init(const int* devices, int n_devices, ecsimd_hip_group** group);
int ecsimd_hip_group_destroy(ecsimd_hip_group* group);
int ecsimd_hip_group_size(const ecsimd_hip_group* group);
int ecsimd_hip_group_uses_rccl(const ecsimd_hip_group* group);                 /* 1 when the gather goes through RCCL */
int ecsimd_hip_group_rccl_version(const ecsimd_hip_group* group);              /* ncclGetVersion of that RCCL (22606 = 2.26.6); 0 without RCCL */
ecsimd_hip_ctx* ecsimd_hip_group_context(ecsimd_hip_group* group, int member); /* e.g. to allocate on that device, set options */
const char* ecsimd_hip_group_last_error(const ecsimd_hip_group* group);
/* curve_group.h:189-218 over the group, device-resident: k[m], x[m], y[m] = member m's slice in ITS device memory;
 * ox, oy, oz = n elements each in member 0's device memory; flags as ecsimd_hip_scalar_mult (| GROUP_NO_GATHER): with
 * OUT_AFFINE oz may be NULL, and oy too (x only: P-256 then runs the ladder without Z on every member).
 * Asynchronous -- except that a call which has to grow a member's staging block (the first one, or a larger batch than
 * any before) drains that member's stream and the gather stream first.  Calls may follow one another without a sync:
 * a member's next ladder is ordered behind the exchange that reads its staging.  ecsimd_hip_group_sync waits for every
 * member and the gather stream and reports how long the last gather took (milliseconds; NULL to skip; -1 if there was none). */
int ecsimd_hip_group_scalar_mult(ecsimd_hip_group* group, int curve, const uint64_t* const* k, const uint64_t* const* x, const uint64_t* const* y,
                                 uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags);
int ecsimd_hip_group_sync(ecsimd_hip_group* group, double* last_gather_ms);
/* Device time (ms, HIP events on that member's stream) of member `member`'s LAST ladder launch; call after group_sync.
 * -1 if the member has not launched anything. */
int ecsimd_hip_group_member_ms(ecsimd_hip_group* group, int member, double* ms);
/* What one GPU can verify of the RCCL side: loads RCCL as group_init does, creates a one-rank communicator on member 0's
 * device and moves `elements` 256-bit elements between two buffers with the calls the gather uses (grouped ncclSend /
 * ncclRecv of ncclUint64, the gather stream).  0 = the bytes arrived intact. */
int ecsimd_hip_group_rccl_selftest(ecsimd_hip_group* group, size_t elements);
/* The same with every array in HOST memory (n elements each): slices copied in, computed, gathered, copied out.
 * Synchronous, and a CONVENIENCE path: the arrays are pageable as far as the library knows, so the runtime stages the
 * copies and they do not overlap the ladders (1.5 GB each way at BASELINE config 4).  Callers that want the overlap keep
 * pinned buffers, device-resident shards and the device form above. */
int ecsimd_hip_group_scalar_mult_host(ecsimd_hip_group* group, int curve, const uint64_t* k, const uint64_t* x, const uint64_t* y,
                                      uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags);

#ifdef __cplusplus
}
#endif
#endif /* ECSIMD_HIP_H */
