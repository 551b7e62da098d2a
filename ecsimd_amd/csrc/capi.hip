// capi.hip -- the C ABI of include/ecsimd_hip.h: argument checks, context/stream handling and
// dispatch to the kernel launchers of kernels.h (gfx950 only; no CPU fallback of any kind).
//
// One field element / one curve point per lane, 256-thread workgroups.  The path is integer-VALU
// bound by three orders of magnitude (~2900 multiplies per byte moved), so there is
// no LDS staging, no XCD-aware remap and no MFMA: workgroups never share data and HBM sees ~0.1 %
// of its bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/ecsimd_hip.h"
#include "kernels.h"
#include "point.cuh"   // curve constants for ecsimd_hip_get_constant (host-side constexpr use only)
#include "gfield.cuh"  // gmod: a run-time modulus as the generic field kernels take it
#include "gcurve.cuh"  // gcurve: a run-time curve as the generic point kernels and the ladder take it

using namespace ecsimd_hip;
using launch::BLOCK;

namespace {
// Stream-ordered store of a 256-bit kernel argument into device memory (the shared scalar of
// scalar_mult_1s): no host buffer has to outlive the call.
__global__ void k_store_words(launch::words8 w, uint32_t* dst) {
  if (threadIdx.x < 8) dst[threadIdx.x] = w.w[threadIdx.x];
}
void store_words(hipStream_t s, const launch::words8& w, uint32_t* dst) {
  hipLaunchKernelGGL(k_store_words, dim3(1), dim3(64), 0, s, w, dst);
}
}  // namespace

// curve dispatch for the per-curve translation units
namespace ecsimd_hip { namespace launch {
// DISPATCH: the kernels every instance has (k_point.inc, k_ladder.inc) -- the two API curves and their
// ECSIMD_HIP_REF_SQUARE_COMPAT twins; DISPATCH2: the kernels of the other algorithms (k_affine.inc, k_varwin.inc), API curves only.
#define DISPATCH(fn, ...) do { switch (curve) { case CURVE_P256: point_launch<CURVE_P256>::fn(__VA_ARGS__); break; case CURVE_SECP256K1: point_launch<CURVE_SECP256K1>::fn(__VA_ARGS__); break; \
    case CURVE_P256_REFSQR: point_launch<CURVE_P256_REFSQR>::fn(__VA_ARGS__); break; default: point_launch<CURVE_SECP256K1_REFSQR>::fn(__VA_ARGS__); break; } } while (0)
#define DISPATCH2(fn, ...) do { if (curve == CURVE_P256) point_launch<CURVE_P256>::fn(__VA_ARGS__); else point_launch<CURVE_SECP256K1>::fn(__VA_ARGS__); } while (0)
void from_affine(hipStream_t s, int curve, const uint64_t* x, const uint64_t* y, uint64_t* jx, uint64_t* jy, uint64_t* jz, size_t n) { DISPATCH(from_affine, s, x, y, jx, jy, jz, n); }
void to_affine(hipStream_t s, int curve, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n) { DISPATCH(to_affine, s, jx, jy, jz, x, y, n); }
void compute_y(hipStream_t s, int curve, const uint64_t* x, uint64_t* y, uint8_t* ok, size_t n) { DISPATCH(compute_y, s, x, y, ok, n); }
void dblu(hipStream_t s, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) { DISPATCH(dblu, s, px, py, pz, rx, ry, rz, n); }
void zaddu(hipStream_t s, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, const uint64_t* qx, const uint64_t* qy, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) { DISPATCH(zaddu, s, px, py, pz, qx, qy, rx, ry, rz, n); }
void zdau(hipStream_t s, int curve, const uint64_t* px, const uint64_t* py, const uint64_t* pz, uint64_t* qx, uint64_t* qy, uint64_t* qz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) { DISPATCH(zdau, s, px, py, pz, qx, qy, qz, rx, ry, rz, n); }
void add_z2_1(hipStream_t s, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) { DISPATCH(add_z2_1, s, ax, ay, az, bx, by, rx, ry, rz, n); }
void trplu(hipStream_t s, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) { DISPATCH(trplu, s, px, py, pz, rx, ry, rz, n); }
void scalar_mult(hipStream_t s, int curve, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) { DISPATCH(scalar_mult, s, k, k_stride, x, y, ox, oy, oz, n, flags); }
void zdau_repeat(hipStream_t s, int curve, const uint64_t* px, const uint64_t* py, const uint64_t* pz, const uint64_t* qx, const uint64_t* qy, uint64_t* rx, uint64_t* ry, uint64_t* sx, uint64_t* sy, uint64_t* oz, size_t n, int iters, uint64_t swap_bits, int radix) { DISPATCH(zdau_repeat, s, px, py, pz, qx, qy, rx, ry, sx, sy, oz, n, iters, swap_bits, radix); }
void to_affine_batched(hipStream_t s, int curve, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n, bool in_fast) { DISPATCH2(to_affine_batched, s, jx, jy, jz, x, y, n, in_fast); }
void pack_table(hipStream_t s, int curve, const uint64_t* tx, const uint64_t* ty, uint32_t* table) { DISPATCH2(pack_table, s, tx, ty, table); }
void pack_table_signed(hipStream_t s, int curve, int wbits, const uint64_t* tx, const uint64_t* ty, uint32_t* table) { DISPATCH2(pack_table_signed, s, wbits, tx, ty, table); }
void base_windowed_signed(hipStream_t s, int curve, int wbits, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, bool constant_time) { DISPATCH2(base_windowed_signed, s, wbits, k, table, ox, oy, oz, n, constant_time); }
void inverse_batched(hipStream_t s, int curve, const uint64_t* a, uint64_t* out, size_t n) { DISPATCH2(inverse_batched, s, a, out, n); }
void x_mod_n_equals(hipStream_t s, int curve, const uint64_t* x, const uint8_t* finite, const uint64_t* r, uint8_t* ok, size_t n) { DISPATCH2(x_mod_n_equals, s, x, finite, r, ok, n); }
void affine_add_batched(hipStream_t s, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n) { DISPATCH2(affine_add_batched, s, ax, ay, bx, by, rx, ry, finite, n); }
void base_windowed(hipStream_t s, int curve, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, bool constant_time) { DISPATCH2(base_windowed, s, k, table, ox, oy, oz, n, constant_time); }
void pack_table_big(hipStream_t s, int curve, const uint64_t* tx, const uint64_t* ty, uint32_t* table) { DISPATCH2(pack_table_big, s, tx, ty, table); }
void base_windowed_big(hipStream_t s, int curve, const uint64_t* k, const uint32_t* table, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n) { DISPATCH2(base_windowed_big, s, k, table, ox, oy, oz, n); }
void add_mixed_complete(hipStream_t s, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) { DISPATCH2(add_mixed_complete, s, ax, ay, az, bx, by, rx, ry, rz, n); }
void varwin_scalar_mult(hipStream_t s, int curve, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, int flags, uint64_t* scratch, uint64_t* ox, uint64_t* oy, size_t n) { DISPATCH2(varwin_scalar_mult, s, k, k_stride, x, y, flags, scratch, ox, oy, n); }
#undef DISPATCH
#undef DISPATCH2
} }

// ================================================================== host side
struct ecsimd_hip_ctx {
  int device;
  hipStream_t own_stream;
  hipStream_t stream;
  hipEvent_t handoff;  // orders a newly selected stream after the work already enqueued on the previous one
  int cus;
  uint32_t* sink;      // 4 KiB scratch: peak-probe sink [0, 1024) and the shared scalar at word 1024-8
  uint32_t* window_table[2];   // per curve: 64 x 16 affine multiples d*16^w*G (built on first use)
  uint32_t* windowct_table[2]; // per curve: the 5- / 6-bit odd-digit table of the constant-time comb (CT_WBITS, CT_WBITS_SECP)
  uint32_t* window6_table[2];  // per curve: signed-window table (SIGNED_WBITS bits): m * 2^(WB i) * G, m = 1..2^(WB-1)
  uint32_t* window16_table[2]; // per curve: signed BIG_WINDOW_BITS-bit windows in device memory (20 bits: 13 x 524 288 entries, 436 MB)
  uint64_t* workspace;         // grow-only scratch for the windowed path's Jacobian intermediates
  size_t workspace_bytes;
  uint64_t* base_special[2];   // per curve: the reference ladder's 3 degenerate scalars and its affine results for them on G (small-batch route of scalar_mult_base)
  uint8_t* valid;              // grow-only: per-lane public-key validity of double_scalar_mult / ecdsa_verify_rx
  size_t valid_bytes;
  int ref_square;              // ecsimd_hip_set_ref_square_compat: the reference's square() as written (mul.h:160-212)
  struct gcomb_entry { int curve; uint32_t* table; uint64_t* special; uint32_t* table7; uint32_t* table5; uint32_t* table20; };      // table7: the signed 7-bit comb (ALG_WINDOWED_SIGNED); table5: the constant-time 5-bit comb; table20: the 20-bit comb in device memory (ALG_WINDOWED_BIG, 436 MB)
  std::vector<gcomb_entry> gcomb;   // per registered curve: the 4-bit odd-digit table of multiples of its generator (k_gcomb.hip) and, for the small-batch route, the ladder's
                                    // three degenerate scalars with its affine results for them on G (as base_special); built on first use
  ecsimd_hip_ctx* helper;      // ecsimd_hip_scalar_mult_host: the second stream's context (created on first use, destroyed with this one)
  uint64_t* hstage; size_t hstage_bytes;    // ... and this context's staging block for one chunk (grow-only)
  char err[256];
};

namespace {

int fail(ecsimd_hip_ctx* ctx, hipError_t e, const char* what) {
  if (ctx) snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, hipGetErrorString(e));
  return ECSIMD_HIP_ERR_HIP;
}
int bad(ecsimd_hip_ctx* ctx, const char* what) {
  if (ctx) snprintf(ctx->err, sizeof ctx->err, "bad argument: %s", what);
  return ECSIMD_HIP_ERR_BAD_ARG;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define REQUIRE_CTX() do { if (!ctx) return ECSIMD_HIP_ERR_BAD_ARG; } while (0)
#define REQUIRE_PTR(p) do { if (!(p) && n) return bad(ctx, #p " is null"); if (!aligned16(p)) return bad(ctx, #p " is not 16-byte aligned"); } while (0)
// the y output of an affine result is optional (x-coordinate only); a Jacobian result needs it
#define REQUIRE_OUT_Y(p) do { if (!(flags & ECSIMD_HIP_OUT_AFFINE)) REQUIRE_PTR(p); else if ((p) && !aligned16(p)) return bad(ctx, #p " is not 16-byte aligned"); } while (0)
#define REQUIRE_CURVE() do { if (curve != ECSIMD_HIP_P256 && curve != ECSIMD_HIP_SECP256K1) return bad(ctx, "unknown curve"); } while (0)

// A curve registered at run time (id >= ECSIMD_HIP_FIRST_REGISTERED_CURVE): the generic kernels of k_gcurve.hip / k_gladder.hip take its record GC.
#define GENERIC_CURVE(call) do { if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) { gcurve GC; if (!lookup_curve(curve, &GC)) return bad(ctx, "unknown curve id"); RUN(call); } } while (0)
// Enqueue one launcher call on the context's stream; report launch errors.
#define RUN(call) do { \
    if (n == 0) return ECSIMD_HIP_OK; \
    if (n > (size_t)0x7fffffff * BLOCK) return bad(ctx, "batch too large"); \
    hipError_t e0_ = hipSetDevice(ctx->device); if (e0_ != hipSuccess) return fail(ctx, e0_, "hipSetDevice"); \
    hipStream_t s = ctx->stream; (void)s; \
    call; \
    hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(ctx, e_, #call); \
    return ECSIMD_HIP_OK; } while (0)

void words_to_limbs(const uint32_t (&w)[8], uint64_t out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

// Grow-only device scratch, ordered by the context's stream (synchronise before switching streams with
// ecsimd_hip_set_stream while a scratch-using call is in flight).  hipMalloc synchronises: callers that
// capture graphs warm the path up once.
bool capturing(ecsimd_hip_ctx* ctx) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}
int ensure_workspace(ecsimd_hip_ctx* ctx, size_t bytes) {
  if (ctx->workspace_bytes >= bytes) return ECSIMD_HIP_OK;
  // Growing frees the old block: pointers a hipGraph captured earlier would dangle, and a capture in progress can neither
  // synchronise nor allocate.  The caller warms the path up at its largest batch BEFORE capturing (include/ecsimd_hip.h).
  if (capturing(ctx)) return bad(ctx, "the context workspace would have to grow during stream capture: run this call once at the largest batch size before capturing");
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess && ctx->workspace) e = hipFree(ctx->workspace);
  ctx->workspace = nullptr; ctx->workspace_bytes = 0;
  if (e == hipSuccess) e = hipMalloc(&ctx->workspace, bytes);
  if (e != hipSuccess) return fail(ctx, e, "workspace hipMalloc");
  ctx->workspace_bytes = bytes;
  return ECSIMD_HIP_OK;
}

int ensure_valid(ecsimd_hip_ctx* ctx, size_t bytes) {
  if (ctx->valid_bytes >= bytes) return ECSIMD_HIP_OK;
  if (capturing(ctx)) return bad(ctx, "the validity buffer would have to grow during stream capture: run this call once at the largest batch size before capturing");
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess && ctx->valid) e = hipFree(ctx->valid);
  ctx->valid = nullptr; ctx->valid_bytes = 0;
  if (e == hipSuccess) e = hipMalloc(&ctx->valid, bytes);
  if (e != hipSuccess) return fail(ctx, e, "validity buffer hipMalloc");
  ctx->valid_bytes = bytes;
  return ECSIMD_HIP_OK;
}

// Window tables, produced with the (parity-checked) ladder kernel itself.
//   bits = 4: 64 x 8 entries (2d + 1) * 16^w * G (odd digits; 64 x 16 entries d * 16^w * G with -DECS_FIXED4_ODD=0);  bits = 6 / 7 (signed windows): 43 x 32 / 37 x 64
//   entries m * 2^(bits i) * G, m = slot + 1.
// ALG_CONSTANT_TIME reads EVERY entry of a window, so the best window is narrower than without it (profiles/r03/ab_constant_time_window_width.txt):
// 5 bits on both curves -- 52 windows x 16 entries, 51 additions, a 53 KB table, THREE 256-thread workgroups per CU (3 waves per SIMD, 146 VGPRs at most,
// no spill): P-256 326.0 M/s, secp256k1 323.0.  Measured against it: 4 bits (k_base_windowed<true>: 64 x 8, 63 additions) 280.7 / 278.1; 6 bits (43 x 32,
// one 1024-thread workgroup per CU) 324.2 / 279.5 (19 spills on secp256k1; at 768 threads 310.8 / 268.4); 7 bits (36 additions, 64 entries to read) 294.1 / 235.2.
#ifndef ECS_CT_P256_BITS
#define ECS_CT_P256_BITS 5          // 4: k_base_windowed<true>; 5 / 6: the signed kernel's template
#endif
#ifndef ECS_CT_SECP_BITS
#define ECS_CT_SECP_BITS 5
#endif
constexpr int CT_WBITS = (ECS_CT_P256_BITS == 5 || ECS_CT_P256_BITS == 6) ? ECS_CT_P256_BITS : 0;
constexpr int CT_WBITS_SECP = (ECS_CT_SECP_BITS == 5 || ECS_CT_SECP_BITS == 6) ? ECS_CT_SECP_BITS : 0;
constexpr int SIGNED_WBITS = 7;     // 37 additions, 151 552 B of LDS (6 -> 43 additions, 88 064 B): measured faster
// The reference ladder's degenerate scalars (curve_group.h:189-218 as written: k forced odd, R0 + R1 = 2^i P): 0 and n end at Z = 0, n - 1,
// 2^256 - n - 1 and 2^256 - n meet n P = infinity inside a formula and return a wrong point.  k < 2^256; nn = the group order.
bool ladder_degenerate(const uint64_t nn[4], const uint64_t k[4]) {
  uint64_t c[4], b = 0;                                        // c = 2^256 - n
  for (int l = 0; l < 4; ++l) { const unsigned __int128 d = (unsigned __int128)0 - nn[l] - b; c[l] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1u; }
  auto eq = [](const uint64_t* a, const uint64_t* v, uint64_t plus) {          // a == v - plus
    uint64_t t[4], bb = plus;
    for (int l = 0; l < 4; ++l) { const unsigned __int128 d = (unsigned __int128)v[l] - bb; t[l] = (uint64_t)d; bb = (uint64_t)(d >> 64) & 1u; }
    return a[0] == t[0] && a[1] == t[1] && a[2] == t[2] && a[3] == t[3];
  };
  const bool zero = (k[0] | k[1] | k[2] | k[3]) == 0;
  return zero || eq(k, nn, 0) || eq(k, nn, 1) || eq(k, c, 0) || eq(k, c, 1);
}
int ensure_window_table(ecsimd_hip_ctx* ctx, int curve, int bits = 4) {
  uint32_t** slot = (bits == 4) ? &ctx->window_table[curve] : (bits == launch::BIG_WINDOW_BITS) ? &ctx->window16_table[curve] : (bits == 5 || bits == 6) ? &ctx->windowct_table[curve] : &ctx->window6_table[curve];
  if (*slot) return ECSIMD_HIP_OK;
  const bool big = (bits == launch::BIG_WINDOW_BITS);        // odd multiples (2d + 1) * 2^(bits w) * G, ceil(256 / bits) windows, no carry window
  const bool odd4 = (bits == 4) && (launch::FIXED4_ENTRIES == 8);        // the 4-bit LDS table with odd digits (kernels.h ECS_FIXED4_ODD)
  const bool odds = !big && bits != 4 && ECS_SIGNED_ODD;                  // the 6- / 7-bit LDS table with odd digits (kernels.h ECS_SIGNED_ODD)
  const int windows = (bits == 4) ? 64 : big ? (256 + bits - 1) / bits : launch::signed_windows(bits), per = (bits == 4) ? launch::FIXED4_ENTRIES : 1 << (bits - 1);
  const size_t table_entries = (size_t)windows * per;
  // odd-digit combs: one more scalar, k*, whose point the kernel substitutes for its own sum (k_affine.inc comb_special: the one
  // odd scalar at which the comb's last mixed addition meets R = T), and a 64-byte record holding k* itself behind it
  const bool odd = big || odd4 || odds;
  const size_t entries = table_entries + (odd ? 1 : 0);
  std::vector<uint64_t> host_k;
  try { host_k.assign(entries * 4, 0); }                       // up to 218 MB of host memory (20-bit windows): nothing may throw across the C ABI
  catch (...) { return bad(ctx, "window table: out of host memory"); }
  uint64_t kstar[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool negate_special = false;                                 // the table's k* entry was computed as (n - k*) G: y -> p - y
  if (odd) {
    // high-to-low accumulation (the 4-bit LDS kernel): k* = n - 2 (n mod 2^bits), and only if bit `bits` of k* is 0 (then its lowest digit
    // is -(n mod 2^bits)); low-to-high (the device-memory table): k* = n - 2 (n mod 2^(bits (windows - 1)))
    uint64_t nn[4], m[4] = {0, 0, 0, 0};
    words_to_limbs(curve == ECSIMD_HIP_P256 ? curve_order<CURVE_P256>::N : curve_order<CURVE_SECP256K1>::N, nn);
    const int low_bits = (big || odds) ? bits * (windows - 1) : bits;       // summed from the bottom (20-bit table, 6- / 7-bit LDS tables) or from the top (4 bits)
    for (int l = 0; l < 4; ++l) {
      const int lo = 64 * l;
      if (low_bits >= lo + 64) m[l] = nn[l];
      else if (low_bits > lo) m[l] = nn[l] & ((1ull << (low_bits - lo)) - 1ull);
    }
    unsigned __int128 acc = 0; uint64_t m2[4];                                 // 2m (m < 2^255)
    for (int l = 0; l < 4; ++l) { acc += (unsigned __int128)m[l] * 2u; m2[l] = (uint64_t)acc; acc >>= 64; }
    uint64_t borrow = 0;
    for (int l = 0; l < 4; ++l) {
      const unsigned __int128 d = (unsigned __int128)nn[l] - m2[l] - borrow;
      kstar[l] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1u;
    }
    if (borrow || (odd4 && ((kstar[bits / 64] >> (bits % 64)) & 1u))) { for (int l = 0; l < 4; ++l) kstar[l] = 0; }   // no such scalar for this kernel
    const bool have = (kstar[0] | kstar[1] | kstar[2] | kstar[3]) != 0;
    for (int l = 0; l < 4; ++l) host_k[table_entries * 4 + l] = have ? kstar[l] : (l == 0 ? 1u : 0u);        // without a k*: any scalar, the point is never used
    // k* G comes from the reference ladder below, and the ladder has degenerate scalars of its own (0, n, n - 1, 2^256 - n - 1, 2^256 - n:
    // DESIGN.md section 5).  The 5-bit comb summed from the bottom has k* = n - 2 (n mod 2^255) = 2^256 - n -- one of them:
    // there the ladder multiplies by n - k* instead (never degenerate when k* is: checked) and the entry's y is negated after the conversion.
    if (have && ladder_degenerate(nn, kstar)) {
      uint64_t alt[4], b2 = 0;
      for (int l = 0; l < 4; ++l) { const unsigned __int128 d = (unsigned __int128)nn[l] - kstar[l] - b2; alt[l] = (uint64_t)d; b2 = (uint64_t)(d >> 64) & 1u; }
      if (b2 || ladder_degenerate(nn, alt)) return bad(ctx, "window table: neither k* nor n - k* is a scalar the ladder multiplies correctly");
      for (int l = 0; l < 4; ++l) host_k[table_entries * 4 + l] = alt[l];
      negate_special = true;
    }
  }
  for (int w = 0; w < windows; ++w)
    for (int d = 0; d < per; ++d) {
      const unsigned mult = odd ? 2u * (unsigned)d + 1u : (bits == 4) ? (unsigned)d : (unsigned)d + 1u;   // multiplier m
      const int pos = bits * w;                                                         // entry = m * 2^pos * G
      uint64_t* e = &host_k[((size_t)w * per + d) * 4];
      const int limb = pos / 64, off = pos % 64;
      if (limb >= 4) continue;                                                          // 16-bit windows: the carry window, m * 2^256 (filled below)
      const unsigned __int128 v = (unsigned __int128)mult << off;
      e[limb] = (uint64_t)v;
      if (limb + 1 < 4) e[limb + 1] = (uint64_t)(v >> 64);
      else if ((uint64_t)(v >> 64) != 0) {
        // m * 2^pos >= 2^256: only the top signed window, where the top digit + carry asks for 2^256 * G.  That
        // point cannot come from the ladder: k = 2^256 mod n is one of its degenerate scalars (the Joye
        // ladder keeps R0 + R1 = 2^i * P, so at i = 256 it meets n * P = infinity; the reference's ladder
        // fails there too).  The slot is filled below by doubling 2^255 * G with the affine-addition kernel.
        for (int l = 0; l < 4; ++l) e[l] = 0;
      }
    }
  // The build needs 6 x entries x 32 B of scratch (1.3 GB for the 20-bit table): a TEMPORARY block freed below, not the
  // context's grow-only workspace -- a caller that verifies one signature must not keep 1.3 GB pinned for it.
  if (capturing(ctx)) return bad(ctx, "a window table would have to be built during stream capture: run this call once before capturing");
  uint64_t* kd = nullptr;
  uint32_t* table = nullptr;
  hipError_t e = hipMalloc(&kd, 6 * entries * 32);
  if (e == hipSuccess) e = hipMalloc(&table, (entries + (odd ? 1 : 0)) * 64);      // odd-digit combs: + the record that holds k*
  uint64_t* tx = kd + entries * 4; uint64_t* ty = tx + entries * 4;
  if (e == hipSuccess) e = hipMemcpyAsync(kd, host_k.data(), entries * 32, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // host_k must outlive the copy
  if (e == hipSuccess) {   // ladder (Jacobian, fast domain) into scratch, then affine classical (x, y)
    uint64_t* jx = ty + entries * 4; uint64_t* jy = jx + entries * 4; uint64_t* jz = jy + entries * 4;
    launch::scalar_mult(ctx->stream, curve, kd, 4, nullptr, nullptr, jx, jy, jz, entries, ECSIMD_HIP_OUT_AFFINE);
    launch::to_affine_batched(ctx->stream, curve, jx, jy, jz, tx, ty, entries, true);
    if (negate_special) launch::field_unop(ctx->stream, curve, launch::F_OPPOSITE, ty + table_entries * 4, ty + table_entries * 4, 1);   // -(x, y) = (x, p - y)
    if (big) {
      launch::pack_table_big(ctx->stream, curve, tx, ty, table);           // odd digits: no carry, no 2^256 * G entry
    } else if (odds) {
      launch::pack_table_signed(ctx->stream, curve, bits, tx, ty, table);  // odd digits: no carry, no 2^256 * G entry
    } else if (bits != 4) {
      // The one reachable entry with m * 2^pos = 2^256 (top digit + carry): the ladder cannot produce 2^256 * G (a degenerate
      // scalar, see above), so it is the entry holding 2^255 * G doubled by the affine-addition kernel.
      uint64_t* sx = ty + entries * 4; uint64_t* sy = sx + 4;                                  // scratch (the Jacobian area is free again)
      const size_t src = ((size_t)(255 / bits) * per + ((size_t)1 << (255 % bits)) - 1) * 4;
      const size_t dst = ((size_t)(256 / bits) * per + ((size_t)1 << (256 % bits)) - 1) * 4;
      launch::affine_add_batched(ctx->stream, curve, tx + src, ty + src, tx + src, ty + src, sx, sy, nullptr, 1);
      e = hipMemcpyAsync(tx + dst, sx, 32, hipMemcpyDeviceToDevice, ctx->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(ty + dst, sy, 32, hipMemcpyDeviceToDevice, ctx->stream);
      if (e == hipSuccess) launch::pack_table_signed(ctx->stream, curve, bits, tx, ty, table);
    } else {
      launch::pack_table(ctx->stream, curve, tx, ty, table);
    }
  }
  if (e == hipSuccess && odd) e = hipMemcpyAsync(table + entries * 16, kstar, 64, hipMemcpyHostToDevice, ctx->stream);   // {k*, 0}: read by comb_special
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipGetLastError();
  (void)hipFree(kd);
  if (e != hipSuccess) { (void)hipFree(table); return fail(ctx, e, "window table build"); }
  *slot = table;
  return ECSIMD_HIP_OK;
}

bool overlaps(const void* a, const void* b) { return a == b; }

// The kernel instance an entry point runs: the API curve, or its reference-square twin (field.cuh) when the
// context option or the call's ECSIMD_HIP_REF_SQUARE_COMPAT flag asks for the reference's bits.
int instance(const ecsimd_hip_ctx* ctx, int curve, int flags = 0) {
  return (ctx->ref_square || (flags & ECSIMD_HIP_REF_SQUARE_COMPAT)) ? curve + (CURVE_P256_REFSQR - CURVE_P256) : curve;
}
static_assert(CURVE_SECP256K1_REFSQR - CURVE_SECP256K1 == CURVE_P256_REFSQR - CURVE_P256, "instance() adds one offset");
#define NO_COMPAT(what) do { if (ctx->ref_square || (flags & ECSIMD_HIP_REF_SQUARE_COMPAT)) return bad(ctx, what " is not the reference's algorithm: no ECSIMD_HIP_REF_SQUARE_COMPAT form"); } while (0)

// The reference ladder, then (for OUT_AFFINE) one simultaneous inversion over the whole batch.
int run_ladder(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y,
               uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  if (n == 0) return ECSIMD_HIP_OK;
  if (n > (size_t)0x7fffffff * BLOCK) return bad(ctx, "batch too large");
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return fail(ctx, e, "hipSetDevice");
  if ((flags & ECSIMD_HIP_OUT_AFFINE) && oy == nullptr && curve == ECSIMD_HIP_P256 && instance(ctx, curve, flags) == curve) {
    // x only: the ladder without Z (point.cuh scalar_mult_ladder_x) -- 8M + 6S per bit instead of 9M + 7S
    int rc = ensure_workspace(ctx, launch::scalar_mult_x_scratch_bytes(n));
    if (rc != ECSIMD_HIP_OK) return rc;
    launch::point_launch<CURVE_P256>::scalar_mult_x(ctx->stream, k, k_stride, x, y, ox, ctx->workspace, n, flags);
    e = hipGetLastError();
    return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "scalar_mult (x only) launch");
  }
  if (flags & ECSIMD_HIP_OUT_AFFINE) {
    int rc = ensure_workspace(ctx, 3 * n * 32);
    if (rc != ECSIMD_HIP_OK) return rc;
    uint64_t* jx = ctx->workspace; uint64_t* jy = jx + 4 * n; uint64_t* jz = jy + 4 * n;
    const int cv = instance(ctx, curve, flags);
    launch::scalar_mult(ctx->stream, cv, k, k_stride, x, y, jx, jy, jz, n, flags);
    // reference-square instances: to_affine() per element with the reference's own power ladder (gfp.h:42-44)
    if (cv != curve) launch::to_affine(ctx->stream, cv, jx, jy, jz, ox, oy, n);
    else launch::to_affine_batched(ctx->stream, curve, jx, jy, jz, ox, oy, n, true);
  } else {
    launch::scalar_mult(ctx->stream, instance(ctx, curve, flags), k, k_stride, x, y, ox, oy, oz, n, flags);
  }
  e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "scalar_mult launch");
}

bool lookup_curve(int id, gcurve* out);          // the curve registry, below
// The same for a curve registered at run time: flags BASE_* | OUT_* | LADDER_RADIX32 | REF_SQUARE_COMPAT; the table-driven ALG_* algorithms exist for the
// two built-in curves only.  x == nullptr: the curve's generator.
int run_gcomb(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, uint64_t* ox, uint64_t* oy, size_t n, int flags);
constexpr int GC_NOT_TAKEN = -1000;
int gc_small_base(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, uint64_t* ox, uint64_t* oy, size_t n, int flags);
int run_gvarwin(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy, size_t n, int flags, size_t reserve);
int run_gladder(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y,
                uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  gcurve GC; if (!lookup_curve(curve, &GC)) return bad(ctx, "unknown curve id");
  if (x == nullptr && k_stride == 4 && (flags & ECSIMD_HIP_ALG_WINDOWED) && !(flags & (ECSIMD_HIP_ALG_WINDOWED_SIGNED | ECSIMD_HIP_ALG_WINDOWED_BIG | ECSIMD_HIP_ALG_NO_ENDOMORPHISM)))
    return run_gcomb(ctx, curve, k, ox, oy, n, flags);          // k G from the generator's table in LDS (k_gcomb.hip)
  if (x == nullptr && k_stride == 4 && (flags & ECSIMD_HIP_ALG_WINDOWED_SIGNED) && !(flags & (ECSIMD_HIP_ALG_WINDOWED | ECSIMD_HIP_ALG_WINDOWED_BIG | ECSIMD_HIP_ALG_NO_ENDOMORPHISM | ECSIMD_HIP_ALG_CONSTANT_TIME)))
    return run_gcomb(ctx, curve, k, ox, oy, n, flags);          // ... from the signed 7-bit comb (public scalars)
  if (x == nullptr && k_stride == 4 && (flags & ECSIMD_HIP_ALG_WINDOWED_BIG) && !(flags & (ECSIMD_HIP_ALG_WINDOWED | ECSIMD_HIP_ALG_WINDOWED_SIGNED | ECSIMD_HIP_ALG_NO_ENDOMORPHISM | ECSIMD_HIP_ALG_CONSTANT_TIME)))
    return run_gcomb(ctx, curve, k, ox, oy, n, flags);          // ... from the 20-bit comb in device memory (public scalars)
  if (x != nullptr && (flags & ECSIMD_HIP_ALG_WINDOWED) && !(flags & (ECSIMD_HIP_ALG_WINDOWED_SIGNED | ECSIMD_HIP_ALG_WINDOWED_BIG | ECSIMD_HIP_ALG_NO_ENDOMORPHISM)))
    return run_gvarwin(ctx, curve, k, k_stride, x, y, ox, oy, n, flags, 0);   // k P from the lane's own table of odd multiples (k_gvarwin.hip); ALG_CONSTANT_TIME: every entry read in every window
  if (flags & (ECSIMD_HIP_ALG_WINDOWED | ECSIMD_HIP_ALG_WINDOWED_SIGNED | ECSIMD_HIP_ALG_WINDOWED_BIG | ECSIMD_HIP_ALG_NO_ENDOMORPHISM | ECSIMD_HIP_ALG_CONSTANT_TIME))
    return bad(ctx, "a registered curve has the reference's ladder, ALG_WINDOWED [| ALG_CONSTANT_TIME] for its generator and for a variable base, and ALG_WINDOWED_SIGNED / ALG_WINDOWED_BIG for its generator: the other combinations (the GLV split, constant-time forms of the signed and big combs) exist for P-256 and secp256k1 or not at all");
  if (n == 0) return ECSIMD_HIP_OK;
  if (n > (size_t)0x7fffffff * BLOCK) return bad(ctx, "batch too large");
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return fail(ctx, e, "hipSetDevice");
  const bool ref = ctx->ref_square || (flags & ECSIMD_HIP_REF_SQUARE_COMPAT);
  const int lf = (flags & (ECSIMD_HIP_BASE_MGRY | ECSIMD_HIP_LADDER_RADIX32)) | (ref ? ECSIMD_HIP_REF_SQUARE_COMPAT : 0);
  if (flags & ECSIMD_HIP_OUT_AFFINE) {
    int rc = ensure_workspace(ctx, 3 * n * 32);
    if (rc != ECSIMD_HIP_OK) return rc;
    uint64_t* jx = ctx->workspace; uint64_t* jy = jx + 4 * n; uint64_t* jz = jy + 4 * n;
    launch::gc_scalar_mult(ctx->stream, GC, k, k_stride, x, y, jx, jy, jz, n, lf);
    if (ref) launch::gc_to_affine(ctx->stream, GC, jx, jy, jz, ox, oy, n, true);       // the reference's own power ladder per element (gfp.h:42-44)
    else launch::gc_to_affine_batched(ctx->stream, GC, jx, jy, jz, ox, oy, n);
  } else {
    launch::gc_scalar_mult(ctx->stream, GC, k, k_stride, x, y, ox, oy, oz, n, lf);
  }
  e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "scalar_mult (registered curve) launch");
}

// Variable-base windowed multiplication in chunks of VARWIN_CHUNK lanes (1 408 B of scratch per lane: the
// per-lane tables live in HBM).  `reserve` bytes at the start of the workspace stay untouched (double_scalar_mult).
constexpr size_t VARWIN_CHUNK = (size_t)1 << 22;
int run_varwin(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y,
               uint64_t* ox, uint64_t* oy, size_t n, int flags, size_t reserve) {
  if (n == 0) return ECSIMD_HIP_OK;
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return fail(ctx, e, "hipSetDevice");
  const size_t chunk = n < VARWIN_CHUNK ? n : VARWIN_CHUNK;
  int rc = ensure_workspace(ctx, reserve + launch::varwin_scratch_bytes(chunk));
  if (rc != ECSIMD_HIP_OK) return rc;
  uint64_t* scratch = ctx->workspace + reserve / 8;
  for (size_t first = 0; first < n; first += chunk) {
    const size_t m = (n - first) < chunk ? (n - first) : chunk;
    launch::varwin_scalar_mult(ctx->stream, curve, k + (size_t)k_stride * first, k_stride, x + 4 * first, y + 4 * first, flags, scratch, ox + 4 * first, oy ? oy + 4 * first : nullptr, m);
  }
  e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "scalar_mult (windowed) launch");
}
}  // namespace

// ---- the small-batch route of scalar_mult_base (see ecsimd_hip_scalar_mult_base)
namespace {
constexpr size_t SMALL_BASE_MAX = (size_t)1 << 16;
// A stream under capture can neither build the comb's table nor the record below: such a call keeps the ladder (and stays capturable).
bool capturing_needs_build(ecsimd_hip_ctx* ctx, int curve) {
  return capturing(ctx) && (!ctx->windowct_table[curve] || !ctx->base_special[curve]);
}
int ensure_base_special(ecsimd_hip_ctx* ctx, int curve) {
  if (ctx->base_special[curve]) return ECSIMD_HIP_OK;
  uint64_t nn[4], host[12];
  words_to_limbs(curve == ECSIMD_HIP_P256 ? curve_order<CURVE_P256>::N : curve_order<CURVE_SECP256K1>::N, nn);
  uint64_t b = 1;                                                // n - 1
  for (int l = 0; l < 4; ++l) { const unsigned __int128 d = (unsigned __int128)nn[l] - b; host[l] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1u; }
  b = 0;                                                         // 2^256 - n
  for (int l = 0; l < 4; ++l) { const unsigned __int128 d = (unsigned __int128)0 - nn[l] - b; host[8 + l] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1u; }
  b = 1;                                                         // 2^256 - n - 1
  for (int l = 0; l < 4; ++l) { const unsigned __int128 d = (unsigned __int128)host[8 + l] - b; host[4 + l] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1u; }
  for (int j = 0; j < 3; ++j) if (!ladder_degenerate(nn, host + 4 * j)) return bad(ctx, "base_special: not a degenerate scalar");
  uint64_t* rec = nullptr; uint64_t* tmp = nullptr;
  hipError_t e = hipMalloc(&rec, 9 * 32);
  if (e == hipSuccess) e = hipMalloc(&tmp, 9 * 32);               // Jacobian scratch of the three ladders
  if (e == hipSuccess) e = hipMemcpyAsync(rec, host, 3 * 32, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);     // `host` is on this stack frame
  if (e == hipSuccess) {
    launch::scalar_mult(ctx->stream, curve, rec, 4, nullptr, nullptr, tmp, tmp + 12, tmp + 24, 3, ECSIMD_HIP_OUT_AFFINE);    // the ladder, fast domain
    launch::to_affine_batched(ctx->stream, curve, tmp, tmp + 12, tmp + 24, rec + 12, rec + 24, 3, true);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  (void)hipFree(tmp);
  if (e != hipSuccess) { (void)hipFree(rec); return fail(ctx, e, "base_special build"); }
  ctx->base_special[curve] = rec;
  return ECSIMD_HIP_OK;
}
}  // namespace

// ================================================================== run-time moduli
// The reference's field layer is generic in the modulus type P (mgry_mul.h:84-121, mgry_csts.h:15-35, gfp.h:17-115): a caller
// instantiates it for any odd 256-bit P at compile time.  Here the same genericity is a run-time registry: a field id names a
// modulus, the host derives what mgry_constants<WBN, P> / mgry_mul_constants derive (R, R^2, -R mod p, m') plus the division-step
// constants, and the generic kernels of k_gfield.hip take the record as a kernel argument.  Ids 0, 1 are the curve primes (their
// special-form kernels, field.cuh); 2, 3 the two group orders; 4... whatever ecsimd_hip_register_modulus was given.  Process-wide,
// append-only, guarded by one mutex (registration is rare; lookups copy 236 bytes).
namespace {
struct u256 { uint64_t l[4]; };
bool u_geq(const u256& a, const u256& b) { for (int i = 3; i >= 0; --i) if (a.l[i] != b.l[i]) return a.l[i] > b.l[i]; return true; }
uint64_t u_sub(u256& r, const u256& a, const u256& b) {
  uint64_t bw = 0;
  for (int i = 0; i < 4; ++i) { const unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - bw; r.l[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1u; }
  return bw;
}
void u_dbl_mod(u256& a, const u256& p) {                       // a < p  ->  2a mod p
  const uint64_t top = a.l[3] >> 63;
  for (int i = 3; i > 0; --i) a.l[i] = (a.l[i] << 1) | (a.l[i - 1] >> 63);
  a.l[0] <<= 1;
  if (top || u_geq(a, p)) { u256 t; (void)u_sub(t, a, p); a = t; }
}
void u_words(uint32_t (&w)[8], const u256& v) { for (int i = 0; i < 4; ++i) { w[2 * i] = (uint32_t)v.l[i]; w[2 * i + 1] = (uint32_t)(v.l[i] >> 32); } }
gmod make_gmod(const uint64_t pl[4], uint32_t flags) {
  gmod M; memset(&M, 0, sizeof M);
  u256 p; for (int i = 0; i < 4; ++i) p.l[i] = pl[i];
  u_words(M.p, p);
  u256 t = {{1, 0, 0, 0}};                                    // 1 < p (p >= 3)
  for (int i = 0; i < 256; ++i) u_dbl_mod(t, p);
  u_words(M.r, t);                                            // R mod p            mgry_csts.h:20 (cbn::div there)
  { u256 nr; (void)u_sub(nr, p, t); u_words(M.negr, nr); }    // -R mod p = p - (R mod p), R mod p in [1, p)   mgry_csts.h:24
  for (int i = 0; i < 256; ++i) u_dbl_mod(t, p);
  u_words(M.rsq, t);                                          // R^2 mod p          mgry_csts.h:21
  for (int i = 0; i < 256; ++i) u_dbl_mod(t, p);
  u_words(M.r3, t);                                           // R^3 mod p
  { u256 two = {{2, 0, 0, 0}}, e; (void)u_sub(e, p, two); u_words(M.pm2, e); }                                     // gfp.h:79-81
  { unsigned __int128 c = 1; u256 q; for (int i = 0; i < 4; ++i) { c += p.l[i]; q.l[i] = (uint64_t)c; c >>= 64; }   // (p + 1) / 4, the carry of p + 1 kept
    u256 h; for (int i = 0; i < 4; ++i) h.l[i] = (q.l[i] >> 2) | ((i < 3 ? q.l[i + 1] : (uint64_t)c) << 62); u_words(M.psqrt, h); }
  for (int i = 0; i < 9; ++i) {                               // p in 30-bit limbs (the top one holds bits 240..255)
    const int bit = 30 * i, limb = bit / 64, off = bit % 64;
    uint64_t v = p.l[limb] >> off;
    if (off > 34 && limb + 1 < 4) v |= p.l[limb + 1] << (64 - off);
    M.p30[i] = (int32_t)(v & (i < 8 ? 0x3fffffffu : 0xffffu));
  }
  uint32_t p0 = (uint32_t)p.l[0], inv = p0;                   // p^-1 mod 2^32 by Newton iteration (mgry_mul.h:33-38 uses cbn::mod_inv)
  for (int i = 0; i < 5; ++i) inv *= 2u - p0 * inv;
  M.pinv30 = inv & 0x3fffffffu;
  M.mprime = 0u - inv;
  M.flags = (flags & GMOD_PRIME) | ((p0 & 3u) == 3u ? GMOD_3MOD4 : 0u);
  return M;
}
struct modulus_registry {
  std::mutex mu;
  std::vector<gmod> mods;                                     // index = field id - 2
  modulus_registry() {
    uint64_t n[4];
    words_to_limbs(curve_order<CURVE_P256>::N, n); mods.push_back(make_gmod(n, GMOD_PRIME));
    words_to_limbs(curve_order<CURVE_SECP256K1>::N, n); mods.push_back(make_gmod(n, GMOD_PRIME));
  }
};
modulus_registry& registry() { static modulus_registry r; return r; }
constexpr int FIRST_FIELD_ID = 2, MAX_FIELDS = 4096;
// the record of field id `id` (>= 2); false if there is none
bool lookup_curve(int id, gcurve* out);
bool lookup_modulus(int id, gmod* out) {
  if (id >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) { gcurve G; if (!lookup_curve(id, &G)) return false; *out = G.F; return true; }     // a registered curve's id names its prime's field
  modulus_registry& r = registry();
  std::lock_guard<std::mutex> g(r.mu);
  if (id < FIRST_FIELD_ID || (size_t)(id - FIRST_FIELD_ID) >= r.mods.size()) return false;
  *out = r.mods[id - FIRST_FIELD_ID];
  return true;
}
// ================================================================== run-time curves (round 5)
// The reference's group layer is a template over ANY curve type -- curve_group<Curve> for every Curve with bn_type, P, A, B, Gx, Gy (curve.h:12-15,
// curve_group.h:20-33) over a field GFp<WBN, P> that needs p = 3 mod 4 (gfp.h:84).  Here a curve is a run-time record with an id >= FIRST_CURVE_ID:
// the host derives what the templates derive (the field's constants as for a modulus, Am = a R, Bm = b R: curve_group.h:31-32) plus the 29-bit-limb
// constants of the ladder's loop, and the generic kernels (k_gcurve.hip, k_gladder.hip) take the 512-byte record as a kernel argument.
constexpr int FIRST_CURVE_ID = ECSIMD_HIP_FIRST_REGISTERED_CURVE, MAX_CURVES = 4096;
bool u_is_zero(const u256& a) { return !(a.l[0] | a.l[1] | a.l[2] | a.l[3]); }
bool u_eq(const u256& a, const u256& b) { return !memcmp(a.l, b.l, 32); }
u256 u_from(const uint64_t v[4]) { u256 r; memcpy(r.l, v, 32); return r; }
u256 u_add_mod(const u256& a, const u256& b, const u256& p) {          // a, b < p
  u256 s; unsigned __int128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.l[i] + b.l[i]; s.l[i] = (uint64_t)c; c >>= 64; }
  if ((uint64_t)c || u_geq(s, p)) { u256 t; (void)u_sub(t, s, p); s = t; }
  return s;
}
u256 u_mul_mod(const u256& a, const u256& b, const u256& p) {          // a, b < p: double-and-add (registration is rare)
  u256 r = {{0, 0, 0, 0}};
  for (int i = 255; i >= 0; --i) { u_dbl_mod(r, p); if ((b.l[i >> 6] >> (i & 63)) & 1u) r = u_add_mod(r, a, p); }
  return r;
}
u256 u_shl_mod(u256 a, int bits, const u256& p) { for (int i = 0; i < bits; ++i) u_dbl_mod(a, p); return a; }
void u_limbs29(int32_t (&out)[9], const u256& v) {                      // tight 29-bit limbs (tools/radix29_model.py to_limbs)
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, limb = bit / 64, off = bit % 64;
    uint64_t w = v.l[limb] >> off;
    if (off > 64 - 29 && limb + 1 < 4) w |= v.l[limb + 1] << (64 - off);
    out[i] = (int32_t)(w & (i < 8 ? 0x1fffffffu : 0xffffffu));
  }
}
struct curve_record { gcurve G; u256 a, b, n; bool has_order; gmod N; bool ecdsa_ok; bool prime_order; };
// What a per-lane window table of a VARIABLE base needs beyond the recoding's n >= 2^255: every point on the curve but infinity has order exactly n, so that
// "no addition inside the loop is exceptional" -- a statement about scalars modulo the ORDER OF THE POINT -- holds for every valid input, not only for
// multiples of the generator.  That is: the group has order n (n divides #E and lies in p's Hasse interval, where no second multiple of an n >= 2^255 fits)
// and n is prime.  Checked once per registration on the host: the Hasse test exactly ((|n - p - 1| / 2 rounded up)^2 <= p; conservative by at most one
// at the interval's edge), primality by Miller-Rabin on 16 fixed bases -- a guard against a wrong parameter set, not against a forged one (the caller vouches
// for p's primality in the same way).
bool u_hasse(const u256& n, const u256& p) {
  u256 p1 = p; { int i = 0; while (i < 4 && ++p1.l[i] == 0) ++i; if (i == 4) return false; }
  u256 d; if (u_geq(n, p1)) (void)u_sub(d, n, p1); else (void)u_sub(d, p1, n);
  if (d.l[3] || d.l[2] > 1) return false;                              // 2 sqrt(p) < 2^129
  unsigned __int128 e = ((unsigned __int128)d.l[1] << 64) | d.l[0];
  e = ((e >> 1) | ((unsigned __int128)d.l[2] << 127)) + (d.l[0] & 1);   // ceil(d / 2) <= 2^128
  if (e == 0 && (d.l[2] | d.l[1] | d.l[0])) return false;              // (it was 2^128: beyond sqrt(p))
  const uint64_t e0 = (uint64_t)e, e1 = (uint64_t)(e >> 64);
  const unsigned __int128 lo = (unsigned __int128)e0 * e0, mid = (unsigned __int128)e0 * e1, hi = (unsigned __int128)e1 * e1;
  u256 sq; unsigned __int128 c;
  sq.l[0] = (uint64_t)lo;
  c = (lo >> 64) + (uint64_t)mid + (uint64_t)mid; sq.l[1] = (uint64_t)c;
  c = (c >> 64) + (mid >> 64) + (mid >> 64) + (uint64_t)hi; sq.l[2] = (uint64_t)c;
  c = (c >> 64) + (hi >> 64); sq.l[3] = (uint64_t)c;
  if (c >> 64) return false;
  return u_geq(p, sq);
}
bool u_probable_prime(const u256& n) {
  if (!(n.l[0] & 1u) || (!(n.l[3] | n.l[2] | n.l[1]) && n.l[0] < 128)) return false;       // (the callers' n is odd and >= 2^255)
  const u256 one = {{1, 0, 0, 0}};
  u256 nm1; (void)u_sub(nm1, n, one);
  int s = 0; u256 d = nm1;
  while (!(d.l[0] & 1u)) { for (int i = 0; i < 3; ++i) d.l[i] = (d.l[i] >> 1) | (d.l[i + 1] << 63); d.l[3] >>= 1; ++s; }
  static const uint64_t bases[16] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53};
  for (uint64_t b : bases) {
    const u256 a = {{b, 0, 0, 0}};
    u256 x = one;
    for (int i = 255; i >= 0; --i) { x = u_mul_mod(x, x, n); if ((d.l[i >> 6] >> (i & 63)) & 1u) x = u_mul_mod(x, a, n); }
    if (u_eq(x, one) || u_eq(x, nm1)) continue;
    bool witness = true;
    for (int r = 1; r < s && witness; ++r) { x = u_mul_mod(x, x, n); if (u_eq(x, nm1)) witness = false; }
    if (witness) return false;
  }
  return true;
}
// the reference ladder's degenerate scalars for a group of order nn (ladder_degenerate above works on uint64_t[4])
bool u_ladder_degenerate(const u256& nn, const u256& k) { return ladder_degenerate(nn.l, k.l); }
struct curve_registry { std::mutex mu; std::vector<curve_record> curves; };
curve_registry& curves() { static curve_registry r; return r; }
bool lookup_curve(int id, gcurve* out) {
  curve_registry& r = curves();
  std::lock_guard<std::mutex> g(r.mu);
  if (id < FIRST_CURVE_ID || (size_t)(id - FIRST_CURVE_ID) >= r.curves.size()) return false;
  *out = r.curves[id - FIRST_CURVE_ID].G;
  return true;
}
bool lookup_curve_record(int id, curve_record* out) {
  curve_registry& r = curves();
  std::lock_guard<std::mutex> g(r.mu);
  if (id < FIRST_CURVE_ID || (size_t)(id - FIRST_CURVE_ID) >= r.curves.size()) return false;
  *out = r.curves[id - FIRST_CURVE_ID];
  return true;
}
}  // namespace

extern "C" {

const char* ecsimd_hip_version(void) { return "ecsimd-hip 0.1 (gfx950)"; }

int ecsimd_hip_init(int device, ecsimd_hip_ctx** out) {
  if (!out) return ECSIMD_HIP_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ECSIMD_HIP_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ECSIMD_HIP_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ECSIMD_HIP_ERR_NO_DEVICE;   // the code object is gfx950-only
  ecsimd_hip_ctx* ctx = new (std::nothrow) ecsimd_hip_ctx();
  if (!ctx) return ECSIMD_HIP_ERR_HIP;
  ctx->device = device; ctx->cus = prop.multiProcessorCount; ctx->err[0] = 0; ctx->sink = nullptr;
  ctx->window_table[0] = ctx->window_table[1] = nullptr; ctx->window6_table[0] = ctx->window6_table[1] = nullptr; ctx->windowct_table[0] = ctx->windowct_table[1] = nullptr; ctx->window16_table[0] = ctx->window16_table[1] = nullptr; ctx->workspace = nullptr; ctx->workspace_bytes = 0; ctx->ref_square = 0; ctx->valid = nullptr; ctx->valid_bytes = 0; ctx->base_special[0] = ctx->base_special[1] = nullptr; ctx->helper = nullptr; ctx->hstage = nullptr; ctx->hstage_bytes = 0;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return ECSIMD_HIP_ERR_HIP; }
  ctx->stream = ctx->own_stream;
  if (hipEventCreateWithFlags(&ctx->handoff, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(ctx->own_stream); delete ctx; return ECSIMD_HIP_ERR_HIP; }
  if (hipMalloc(&ctx->sink, 4096) != hipSuccess) { (void)hipEventDestroy(ctx->handoff); (void)hipStreamDestroy(ctx->own_stream); delete ctx; return ECSIMD_HIP_ERR_HIP; }
  *out = ctx;
  return ECSIMD_HIP_OK;
}
int ecsimd_hip_destroy(ecsimd_hip_ctx* ctx) {
  REQUIRE_CTX();
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(ctx->sink);
  (void)hipFree(ctx->window_table[0]); (void)hipFree(ctx->window_table[1]); (void)hipFree(ctx->window6_table[0]); (void)hipFree(ctx->window6_table[1]); (void)hipFree(ctx->windowct_table[0]); (void)hipFree(ctx->windowct_table[1]); (void)hipFree(ctx->window16_table[0]); (void)hipFree(ctx->window16_table[1]); (void)hipFree(ctx->workspace); (void)hipFree(ctx->valid); (void)hipFree(ctx->base_special[0]); (void)hipFree(ctx->base_special[1]);
  for (auto& t : ctx->gcomb) { (void)hipFree(t.table); (void)hipFree(t.special); (void)hipFree(t.table7); (void)hipFree(t.table5); (void)hipFree(t.table20); }
  (void)hipFree(ctx->hstage);
  if (ctx->helper) (void)ecsimd_hip_destroy(ctx->helper);
  (void)hipEventDestroy(ctx->handoff);
  (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return ECSIMD_HIP_OK;
}
// The context's scratch (workspace, the shared-scalar slot, the lazily built tables) is ordered by its stream.
// Selecting another stream therefore makes that stream wait for what this context already enqueued on the
// previous one (an event, no host synchronisation); re-selecting the current stream costs nothing.
static int switch_stream(ecsimd_hip_ctx* ctx, hipStream_t next) {
  if (next == ctx->stream) return ECSIMD_HIP_OK;
  (void)hipSetDevice(ctx->device);
  // A stream under graph capture can neither wait for an outside event nor lend one: there the caller
  // brackets the capture itself (synchronise before capturing, as graph users do for every library).
  hipStreamCaptureStatus from = hipStreamCaptureStatusNone, to = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(ctx->stream, &from); (void)hipStreamIsCapturing(next, &to);
  (void)hipGetLastError();
  if (from != hipStreamCaptureStatusNone || to != hipStreamCaptureStatusNone) { ctx->stream = next; return ECSIMD_HIP_OK; }
  hipError_t e = hipEventRecord(ctx->handoff, ctx->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(next, ctx->handoff, 0);
  if (e != hipSuccess) return fail(ctx, e, "stream hand-off");
  ctx->stream = next;
  return ECSIMD_HIP_OK;
}
int ecsimd_hip_set_stream(ecsimd_hip_ctx* ctx, void* s) { REQUIRE_CTX(); return switch_stream(ctx, (hipStream_t)s); }
int ecsimd_hip_use_own_stream(ecsimd_hip_ctx* ctx) { REQUIRE_CTX(); return switch_stream(ctx, ctx->own_stream); }
int ecsimd_hip_set_ref_square_compat(ecsimd_hip_ctx* ctx, int on) { REQUIRE_CTX(); ctx->ref_square = on ? 1 : 0; return ECSIMD_HIP_OK; }
int ecsimd_hip_get_ref_square_compat(const ecsimd_hip_ctx* ctx) { return ctx ? ctx->ref_square : ECSIMD_HIP_ERR_BAD_ARG; }
int ecsimd_hip_sync(ecsimd_hip_ctx* ctx) {
  REQUIRE_CTX();
  hipError_t e = hipStreamSynchronize(ctx->stream);
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "hipStreamSynchronize");
}
const char* ecsimd_hip_last_error(const ecsimd_hip_ctx* ctx) { return ctx ? ctx->err : "null context"; }
int ecsimd_hip_malloc(ecsimd_hip_ctx* ctx, void** p, size_t bytes) {
  REQUIRE_CTX(); if (!p) return bad(ctx, "dptr is null");
  (void)hipSetDevice(ctx->device);
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "hipMalloc");
}
int ecsimd_hip_free(ecsimd_hip_ctx* ctx, void* p) {
  REQUIRE_CTX(); (void)hipSetDevice(ctx->device);
  hipError_t e = hipFree(p);
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "hipFree");
}
int ecsimd_hip_memcpy_h2d(ecsimd_hip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  REQUIRE_CTX(); (void)hipSetDevice(ctx->device);
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);     // src may be pageable: complete before returning
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "hipMemcpy h2d");
}
int ecsimd_hip_memcpy_d2h(ecsimd_hip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  REQUIRE_CTX(); (void)hipSetDevice(ctx->device);
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "hipMemcpy d2h");
}
int ecsimd_hip_memcpy_d2d(ecsimd_hip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  REQUIRE_CTX(); (void)hipSetDevice(ctx->device);
  if (bytes == 0) return ECSIMD_HIP_OK;
  if (!dst || !src) return bad(ctx, "memcpy_d2d: null pointer");
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream);
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "hipMemcpy d2d");
}


int ecsimd_hip_register_modulus(const uint64_t p[4], int flags, int* field_id) {
  if (!p || !field_id) return ECSIMD_HIP_ERR_BAD_ARG;
  if (!(p[0] & 1u) || (p[0] < 3 && !(p[1] | p[2] | p[3]))) return ECSIMD_HIP_ERR_BAD_ARG;        // odd, >= 3 (Montgomery arithmetic needs gcd(p, 2^256) = 1)
  if (flags & ~ECSIMD_HIP_MODULUS_PRIME) return ECSIMD_HIP_ERR_BAD_ARG;
  // the curve primes keep their special-form kernels
  { uint64_t c[4]; words_to_limbs(curve_consts<CURVE_P256>::P, c); if (!memcmp(c, p, 32)) { *field_id = ECSIMD_HIP_P256; return ECSIMD_HIP_OK; }
    words_to_limbs(curve_consts<CURVE_SECP256K1>::P, c); if (!memcmp(c, p, 32)) { *field_id = ECSIMD_HIP_SECP256K1; return ECSIMD_HIP_OK; } }
  try {
    const gmod M = make_gmod(p, (flags & ECSIMD_HIP_MODULUS_PRIME) ? GMOD_PRIME : 0u);
    modulus_registry& r = registry();
    std::lock_guard<std::mutex> g(r.mu);
    // Entries are keyed on (p, PRIME): the flag changes what gfp_inverse computes (the shared division-step inversion against gfp.h:42-44's x^(p-2), which
    // differ for a composite p), so a registration with the flag must never change the behaviour of an id someone else holds without it.
    for (size_t i = 0; i < r.mods.size(); ++i)
      if (!memcmp(r.mods[i].p, M.p, sizeof M.p) && (r.mods[i].flags == M.flags || i < 2)) { *field_id = FIRST_FIELD_ID + (int)i; return ECSIMD_HIP_OK; }   // (i < 2: the built-in group orders ARE prime, whatever the caller says)
    if (r.mods.size() >= (size_t)MAX_FIELDS) return ECSIMD_HIP_ERR_BAD_ARG;
    r.mods.push_back(M);
    *field_id = FIRST_FIELD_ID + (int)r.mods.size() - 1;
    return ECSIMD_HIP_OK;
  } catch (...) { return ECSIMD_HIP_ERR_BAD_ARG; }             // nothing throws across the C ABI
}

// curve_group<Curve> for any Curve (curve.h:12-15): what the templates derive at compile time, derived here once per curve.
int ecsimd_hip_register_curve(const uint64_t p[4], const uint64_t a[4], const uint64_t b[4], const uint64_t gx[4], const uint64_t gy[4], const uint64_t n[4], int flags, int* curve_id) {
  if (!p || !a || !b || !gx || !gy || !curve_id) return ECSIMD_HIP_ERR_BAD_ARG;
  if (flags & ~ECSIMD_HIP_CURVE_GENERIC_KERNELS) return ECSIMD_HIP_ERR_BAD_ARG;
  const u256 P = u_from(p), A = u_from(a), B = u_from(b), GX = u_from(gx), GY = u_from(gy);
  if ((p[0] & 3u) != 3u || (p[0] < 7 && !(p[1] | p[2] | p[3]))) return ECSIMD_HIP_ERR_BAD_ARG;     // p = 3 mod 4 like the reference's GFp (gfp.h:84), p >= 7
  if (u_geq(A, P) || u_geq(B, P) || u_geq(GX, P) || u_geq(GY, P)) return ECSIMD_HIP_ERR_BAD_ARG;
  // the generator is on the curve (a typo in any of the five values is caught here): y^2 = x^3 + a x + b
  { const u256 lhs = u_mul_mod(GY, GY, P), x3 = u_mul_mod(u_mul_mod(GX, GX, P), GX, P);
    if (!u_eq(lhs, u_add_mod(u_add_mod(x3, u_mul_mod(A, GX, P), P), B, P))) return ECSIMD_HIP_ERR_BAD_ARG; }
  // non-singular: 4 a^3 + 27 b^2 != 0
  { const u256 a3 = u_mul_mod(u_mul_mod(A, A, P), A, P), b2 = u_mul_mod(B, B, P);
    u256 four = {{4, 0, 0, 0}}, c27 = {{27, 0, 0, 0}};
    if (u_geq(four, P)) (void)u_sub(four, four, P);
    while (u_geq(c27, P)) (void)u_sub(c27, c27, P);
    if (u_is_zero(u_add_mod(u_mul_mod(four, a3, P), u_mul_mod(c27, b2, P), P))) return ECSIMD_HIP_ERR_BAD_ARG; }
  u256 N = {{0, 0, 0, 0}};
  if (n) { N = u_from(n); if (!(n[0] & 1u) || u_is_zero(N)) return ECSIMD_HIP_ERR_BAD_ARG; }
  if (!(flags & ECSIMD_HIP_CURVE_GENERIC_KERNELS)) {               // the two built-in curves keep their special-form kernels
    for (int cv = 0; cv < 2; ++cv) {
      uint64_t c[5][4];
      for (int w = 0; w < 5; ++w) (void)ecsimd_hip_get_constant(cv, w, c[w]);
      if (!memcmp(c[0], p, 32) && !memcmp(c[1], a, 32) && !memcmp(c[2], b, 32) && !memcmp(c[3], gx, 32) && !memcmp(c[4], gy, 32)) { *curve_id = cv; return ECSIMD_HIP_OK; }
    }
  }
  try {
    curve_record rec; memset(&rec, 0, sizeof rec);
    rec.a = A; rec.b = B; rec.n = N; rec.has_order = n != nullptr;
    rec.ecdsa_ok = false;
    if (n) {
      // ECDSA on this curve multiplies through the reference's ladder, which is wrong at n - 1, 2^256 - n - 1 and 2^256 - n: such a scalar u is replaced by n - u
      // (k_gc_ladder_safe_scalars).  That needs n - u to be a good scalar in turn; and x mod n by ONE conditional subtraction needs p < 2n.
      rec.N = make_gmod(n, GMOD_PRIME);
      u256 one = {{1, 0, 0, 0}}, zero = {{0, 0, 0, 0}}, nm1, c, cm1, alt, twice;
      (void)u_sub(nm1, N, one); (void)u_sub(c, zero, N); (void)u_sub(cm1, c, one);
      bool safe = true;
      for (const u256* d : {&nm1, &c, &cm1}) if (!u_geq(*d, N)) { (void)u_sub(alt, N, *d); if (u_is_zero(alt) || u_ladder_degenerate(N, alt)) safe = false; }
      const uint64_t top = N.l[3] >> 63;
      for (int i = 3; i > 0; --i) twice.l[i] = (N.l[i] << 1) | (N.l[i - 1] >> 63);
      twice.l[0] = N.l[0] << 1;
      const bool p_below_2n = top || !u_geq(P, twice);
      rec.ecdsa_ok = safe && p_below_2n;
    }
    gcurve& G = rec.G;
    G.F = make_gmod(p, GMOD_PRIME);                                 // registering a CURVE vouches that p is prime
    u_words(G.am, u_shl_mod(A, 256, P));                            // to_mgry(A), to_mgry(B): curve_group.h:31-32
    u_words(G.bm, u_shl_mod(B, 256, P));
    u_words(G.gx, GX); u_words(G.gy, GY);
    u_limbs29(G.r29.p, P);
    G.r29.qinv = G.F.mprime & 0x1fffffffu;                          // -p^-1 mod 2^29 (mprime = -p^-1 mod 2^32)
    const u256 one = {{1, 0, 0, 0}};
    u_limbs29(G.r29.in, u_shl_mod(one, 266, P));                    // 2^266 mod p: x 2^256 -> x 2^261 through one product / 2^261
    u_limbs29(G.r29.out, u_shl_mod(one, 256, P));                   // 2^256 mod p: back
    curve_registry& r = curves();
    std::lock_guard<std::mutex> g(r.mu);
    // Entries are keyed on the curve AND the order it was registered with (none is a value of its own): what an id does -- the small-batch route of
    // scalar_mult_base, the comb, ECDSA -- depends on n, so a later registration with another n (or the first one with any) must never change the
    // behaviour of an id somebody else holds (the rule the modulus registry follows for its PRIME flag).
    for (size_t i = 0; i < r.curves.size(); ++i) {
      const curve_record& o = r.curves[i];
      if (!memcmp(o.G.F.p, G.F.p, 32) && u_eq(o.a, A) && u_eq(o.b, B) && !memcmp(o.G.gx, G.gx, 32) && !memcmp(o.G.gy, G.gy, 32) &&
          o.has_order == rec.has_order && (!rec.has_order || u_eq(o.n, N))) { *curve_id = FIRST_CURVE_ID + (int)i; return ECSIMD_HIP_OK; }
    }
    if (r.curves.size() >= (size_t)MAX_CURVES) return ECSIMD_HIP_ERR_BAD_ARG;
    rec.prime_order = rec.has_order && (N.l[3] >> 63) != 0 && u_hasse(N, P) && u_probable_prime(N);   // (a new record only: ~0.1 s of host arithmetic)
    r.curves.push_back(rec);
    *curve_id = FIRST_CURVE_ID + (int)r.curves.size() - 1;
    return ECSIMD_HIP_OK;
  } catch (...) { return ECSIMD_HIP_ERR_BAD_ARG; }
}

int ecsimd_hip_curve_capabilities(int curve, int* caps) {
  if (!caps) return ECSIMD_HIP_ERR_BAD_ARG;
  const int all = ECSIMD_HIP_CURVE_HAS_ORDER | ECSIMD_HIP_CURVE_COMB | ECSIMD_HIP_CURVE_ECDSA | ECSIMD_HIP_CURVE_WINDOW_VARIABLE_BASE;
  if (curve == ECSIMD_HIP_P256 || curve == ECSIMD_HIP_SECP256K1) { *caps = all; return ECSIMD_HIP_OK; }
  curve_record rec; if (!lookup_curve_record(curve, &rec)) return ECSIMD_HIP_ERR_BAD_ARG;
  const bool comb = rec.has_order && (rec.n.l[3] >> 63) != 0;
  *caps = (rec.has_order ? ECSIMD_HIP_CURVE_HAS_ORDER : 0) | (comb ? ECSIMD_HIP_CURVE_COMB : 0) | (rec.has_order && rec.ecdsa_ok ? ECSIMD_HIP_CURVE_ECDSA : 0) |
          (comb && rec.prime_order ? ECSIMD_HIP_CURVE_WINDOW_VARIABLE_BASE : 0);
  return ECSIMD_HIP_OK;
}

int ecsimd_hip_get_constant(int curve, int which, uint64_t out[4]) {
  if (!out || which < 0 || which > 11) return ECSIMD_HIP_ERR_BAD_ARG;
#define PICK(C) do { using K = curve_consts<C>; using E = curve_exps<C>; \
    switch (which) { \
      case 0: words_to_limbs(K::P, out); break; \
      case 1: { uint32_t a[8]; for (int i = 0; i < 8; ++i) a[i] = (C == CURVE_P256) ? K::P[i] : 0u; if (C == CURVE_P256) a[0] -= 3u; words_to_limbs(a, out); break; } \
      case 2: { static const uint32_t b256[8] = {0x27d2604bu, 0x3bce3c3eu, 0xcc53b0f6u, 0x651d06b0u, 0x769886bcu, 0xb3ebbd55u, 0xaa3a93e7u, 0x5ac635d8u}; \
                static const uint32_t bk1[8] = {7u, 0, 0, 0, 0, 0, 0, 0}; words_to_limbs((C == CURVE_P256) ? b256 : bk1, out); break; } \
      case 3: words_to_limbs(K::GX, out); break; \
      case 4: words_to_limbs(K::GY, out); break; \
      case 5: words_to_limbs(K::R_P, out); break; \
      case 6: words_to_limbs(K::RSQ, out); break; \
      case 7: { /* -R mod p = p - (R mod p) */ uint64_t p[4], r[4]; words_to_limbs(K::P, p); words_to_limbs(K::R_P, r); \
                unsigned __int128 bw = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)p[i] - r[i] - (uint64_t)bw; out[i] = (uint64_t)d; bw = (d >> 64) & 1; } break; } \
      case 8: words_to_limbs(K::AM, out); break; \
      case 9: words_to_limbs(K::BM, out); break; \
      case 10: words_to_limbs(E::P_M2, out); break; \
      default: words_to_limbs(E::P_SQRT, out); break; \
    } } while (0)
  if (curve == ECSIMD_HIP_P256) PICK(CURVE_P256);
  else if (curve == ECSIMD_HIP_SECP256K1) PICK(CURVE_SECP256K1);
  else {
    // a registered curve: every slot
    curve_record rec;
    if (lookup_curve_record(curve, &rec)) {
      switch (which) {
        case 1: memcpy(out, rec.a.l, 32); return ECSIMD_HIP_OK;
        case 2: memcpy(out, rec.b.l, 32); return ECSIMD_HIP_OK;
        case 3: words_to_limbs(rec.G.gx, out); return ECSIMD_HIP_OK;
        case 4: words_to_limbs(rec.G.gy, out); return ECSIMD_HIP_OK;
        case 8: words_to_limbs(rec.G.am, out); return ECSIMD_HIP_OK;
        case 9: words_to_limbs(rec.G.bm, out); return ECSIMD_HIP_OK;
        default: break;                                    // the field's slots below
      }
    }
    // a field id: p, R mod p, R^2 mod p, -R mod p, p - 2, (p + 1) / 4; the curve slots (a, b, Gx, Gy, a R, b R) read zero
    gmod M;
    if (!lookup_modulus(curve, &M)) return ECSIMD_HIP_ERR_BAD_ARG;
    static const uint32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    switch (which) {
      case 0: words_to_limbs(M.p, out); break;
      case 5: words_to_limbs(M.r, out); break;
      case 6: words_to_limbs(M.rsq, out); break;
      case 7: words_to_limbs(M.negr, out); break;
      case 10: words_to_limbs(M.pm2, out); break;
      case 11: words_to_limbs(M.psqrt, out); break;
      default: words_to_limbs(zero, out); break;
    }
  }
#undef PICK
  return ECSIMD_HIP_OK;
}

// ---- L2
int ecsimd_hip_add(ecsimd_hip_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* carry, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); REQUIRE_PTR(out); RUN(launch::add(s, a, b, out, carry, n)); }
int ecsimd_hip_sub(ecsimd_hip_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* borrow, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); REQUIRE_PTR(out); RUN(launch::sub(s, a, b, out, borrow, n)); }
int ecsimd_hip_sub_if_above(ecsimd_hip_ctx* ctx, const uint64_t* a, const uint64_t* p, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(p); REQUIRE_PTR(out); RUN(launch::sub_if_above(s, a, p, out, n)); }
int ecsimd_hip_cmp_lt(ecsimd_hip_ctx* ctx, const uint64_t* a, const uint64_t* b, uint8_t* flag, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); if (!flag && n) return bad(ctx, "flag is null");
  RUN(launch::sub(s, a, b, nullptr, flag, n)); }
int ecsimd_hip_cmp_eq(ecsimd_hip_ctx* ctx, const uint64_t* a, const uint64_t* b, int limbs, uint8_t* flag, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); if (!flag && n) return bad(ctx, "flag is null");
  if (limbs < 1 || limbs > 8) return bad(ctx, "limbs must be 1..8");
  RUN(launch::cmp_eq(s, a, b, limbs, flag, n)); }
int ecsimd_hip_mask_op(ecsimd_hip_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  REQUIRE_CTX(); if (op < ECSIMD_HIP_MASK_NOT || op > ECSIMD_HIP_MASK_EQ) return bad(ctx, "unknown mask operation");
  if (n && (!a || !out || (op != ECSIMD_HIP_MASK_NOT && !b))) return bad(ctx, "mask pointer is null");
  RUN(launch::mask_op(s, op, a, b, out, n)); }
int ecsimd_hip_mask_count(ecsimd_hip_ctx* ctx, const uint8_t* a, size_t n, size_t* count) {
  REQUIRE_CTX(); if (!count) return bad(ctx, "count is null");
  *count = 0;
  if (n == 0) return ECSIMD_HIP_OK;
  if (!a) return bad(ctx, "mask pointer is null");
  (void)hipSetDevice(ctx->device);
  unsigned long long* slot = reinterpret_cast<unsigned long long*>(ctx->sink + 1024 - 16);     // 8-byte slot next to the shared scalar
  hipError_t e = hipMemsetAsync(slot, 0, sizeof *slot, ctx->stream);
  if (e == hipSuccess) { launch::mask_count(ctx->stream, a, n, slot); e = hipGetLastError(); }
  unsigned long long host = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&host, slot, sizeof host, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail(ctx, e, "mask_count");
  *count = (size_t)host;
  return ECSIMD_HIP_OK; }
int ecsimd_hip_shift_left_one(ecsimd_hip_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* carry, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out); RUN(launch::shift_left_one(s, a, out, carry, n)); }
int ecsimd_hip_mul(ecsimd_hip_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out8, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); REQUIRE_PTR(out8); RUN(launch::mul(s, a, b, out8, n)); }
int ecsimd_hip_square(ecsimd_hip_ctx* ctx, const uint64_t* a, uint64_t* out8, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out8); RUN(launch::square(s, a, out8, n, ctx->ref_square != 0)); }
int ecsimd_hip_swap_if(ecsimd_hip_ctx* ctx, const uint8_t* mask, uint64_t* a, uint64_t* b, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); if (!mask && n) return bad(ctx, "mask is null"); RUN(launch::swap_if(s, mask, a, b, n)); }
int ecsimd_hip_if_else(ecsimd_hip_ctx* ctx, const uint8_t* mask, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); REQUIRE_PTR(out); if (!mask && n) return bad(ctx, "mask is null"); RUN(launch::if_else(s, mask, a, b, out, n)); }

// ---- wire formats
int ecsimd_hip_from_bytes_be(ecsimd_hip_ctx* ctx, const uint8_t* bytes, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(bytes); REQUIRE_PTR(out); RUN(launch::bytes_be(s, bytes, out, n)); }
int ecsimd_hip_to_bytes_be(ecsimd_hip_ctx* ctx, const uint64_t* in, uint8_t* bytes, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(in); REQUIRE_PTR(bytes); RUN(launch::bytes_be(s, in, bytes, n)); }
// The reference's register layout (four lanes per wide, limb-major: bignum.h:99-100) <-> the ABI's one element per lane, on the device.
int ecsimd_hip_wide4_to_lanes(ecsimd_hip_ctx* ctx, const void* wides, size_t record_bytes, size_t offset_bytes, uint64_t* out, size_t n_wides) {
  REQUIRE_CTX(); const size_t n = 4 * n_wides; REQUIRE_PTR(out); if (!wides && n) return bad(ctx, "wides is null");
  if (static_cast<const void*>(out) == wides) return bad(ctx, "the transposition is not in place: out must not alias wides");
  if (record_bytes < 128 || (record_bytes & 7u) || (offset_bytes & 7u) || offset_bytes + 128 > record_bytes || (reinterpret_cast<uintptr_t>(wides) & 7u)) return bad(ctx, "a wide is 128 bytes at an 8-byte aligned offset inside its record");
  if (n_wides > (size_t)0x7fffffff * (BLOCK / 4)) return bad(ctx, "batch too large");
  RUN(launch::wide4_to_lanes(s, wides, record_bytes, offset_bytes, out, n)); }
int ecsimd_hip_lanes_to_wide4(ecsimd_hip_ctx* ctx, const uint64_t* in, void* wides, size_t record_bytes, size_t offset_bytes, size_t n_wides) {
  REQUIRE_CTX(); const size_t n = 4 * n_wides; REQUIRE_PTR(in); if (!wides && n) return bad(ctx, "wides is null");
  if (static_cast<const void*>(in) == wides) return bad(ctx, "the transposition is not in place: wides must not alias in");
  if (record_bytes < 128 || (record_bytes & 7u) || (offset_bytes & 7u) || offset_bytes + 128 > record_bytes || (reinterpret_cast<uintptr_t>(wides) & 7u)) return bad(ctx, "a wide is 128 bytes at an 8-byte aligned offset inside its record");
  if (n_wides > (size_t)0x7fffffff * (BLOCK / 4)) return bad(ctx, "batch too large");
  RUN(launch::lanes_to_wide4(s, in, wides, record_bytes, offset_bytes, n)); }
int ecsimd_hip_mask_bit(ecsimd_hip_ctx* ctx, const uint64_t* a, int bit, uint8_t* flag, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); if (!flag && n) return bad(ctx, "flag is null"); if (bit < 0 || bit > 255) return bad(ctx, "bit index");
  RUN(launch::mask_bit(s, a, bit, flag, n)); }
int ecsimd_hip_sec1_encode(ecsimd_hip_ctx* ctx, int curve, const uint64_t* x, const uint64_t* y, uint8_t* out, size_t n, int compressed) {
  REQUIRE_CTX(); REQUIRE_PTR(x); REQUIRE_PTR(y); REQUIRE_PTR(out);
  if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) { gcurve GC; if (!lookup_curve(curve, &GC)) return bad(ctx, "unknown curve id"); curve = ECSIMD_HIP_P256; }     // the encoding does not look at the curve
  REQUIRE_CURVE(); RUN(launch::sec1_encode(s, curve, x, y, out, n, compressed != 0)); }
int ecsimd_hip_sec1_decode(ecsimd_hip_ctx* ctx, int curve, const uint8_t* in, uint64_t* x, uint64_t* y, uint8_t* ok, size_t n, int compressed) {
  REQUIRE_CTX(); REQUIRE_PTR(in); REQUIRE_PTR(x); REQUIRE_PTR(y); GENERIC_CURVE(launch::gc_sec1_decode(s, GC, in, x, y, ok, n, compressed != 0));
  REQUIRE_CURVE(); RUN(launch::sec1_decode(s, curve, in, x, y, ok, n, compressed != 0)); }

// ---- L3.  `curve` is a FIELD id here: a curve's prime (0, 1: the special-form kernels of field.cuh) or a run-time modulus (>= 2: k_gfield.hip).
#define FIELD_OR_CURVE(generic_call) do { if (curve != ECSIMD_HIP_P256 && curve != ECSIMD_HIP_SECP256K1) { \
    gmod M; if (!lookup_modulus(curve, &M)) return bad(ctx, "unknown curve / field id"); RUN(generic_call); } } while (0)
int ecsimd_hip_mod_add(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); REQUIRE_PTR(out); FIELD_OR_CURVE(launch::gfield_binop(s, M, launch::F_MOD_ADD, a, b, out, n)); RUN(launch::field_binop(s, instance(ctx, curve), launch::F_MOD_ADD, a, b, out, n)); }
int ecsimd_hip_mod_sub(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); REQUIRE_PTR(out); FIELD_OR_CURVE(launch::gfield_binop(s, M, launch::F_MOD_SUB, a, b, out, n)); RUN(launch::field_binop(s, instance(ctx, curve), launch::F_MOD_SUB, a, b, out, n)); }
int ecsimd_hip_mod_mul(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); REQUIRE_PTR(out); FIELD_OR_CURVE(launch::gfield_mod_mul(s, M, a, b, out, n)); RUN(launch::mod_mul(s, instance(ctx, curve), a, b, out, n)); }
int ecsimd_hip_mod_shift_left(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, int count, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out); if ((count & 0xff) < 1 || (count & ~0x1ff)) return bad(ctx, "count must be 1..255, optionally | ECSIMD_HIP_SHIFT_FUSED"); FIELD_OR_CURVE(launch::gfield_shift_left(s, M, a, count, out, n)); RUN(launch::mod_shift_left(s, instance(ctx, curve), a, count, out, n)); }
int ecsimd_hip_mgry_reduce(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a8, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a8); REQUIRE_PTR(out); FIELD_OR_CURVE(launch::gfield_reduce(s, M, a8, out, n)); RUN(launch::mgry_reduce(s, instance(ctx, curve), a8, out, n)); }
int ecsimd_hip_mgry_mul(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(b); REQUIRE_PTR(out); FIELD_OR_CURVE(launch::gfield_binop(s, M, launch::F_MGRY_MUL, a, b, out, n)); RUN(launch::field_binop(s, instance(ctx, curve), launch::F_MGRY_MUL, a, b, out, n)); }
int ecsimd_hip_mgry_sqr(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out); FIELD_OR_CURVE(launch::gfield_unop(s, M, launch::F_MGRY_SQR, a, out, n, ctx->ref_square != 0)); RUN(launch::field_unop(s, instance(ctx, curve), launch::F_MGRY_SQR, a, out, n)); }
int ecsimd_hip_mgry_from_classical(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out); FIELD_OR_CURVE(launch::gfield_unop(s, M, launch::F_FROM_CLASSICAL, a, out, n, false)); RUN(launch::field_unop(s, instance(ctx, curve), launch::F_FROM_CLASSICAL, a, out, n)); }
int ecsimd_hip_mgry_to_classical(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out); FIELD_OR_CURVE(launch::gfield_unop(s, M, launch::F_TO_CLASSICAL, a, out, n, false)); RUN(launch::field_unop(s, instance(ctx, curve), launch::F_TO_CLASSICAL, a, out, n)); }
int ecsimd_hip_mgry_pow(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, const uint64_t exponent[4], uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out); if (!exponent) return bad(ctx, "exponent is null");
  launch::words8 e; for (int i = 0; i < 4; ++i) { e.w[2 * i] = (uint32_t)exponent[i]; e.w[2 * i + 1] = (uint32_t)(exponent[i] >> 32); }
  FIELD_OR_CURVE(launch::gfield_pow(s, M, a, e, out, n, ctx->ref_square != 0));
  RUN(launch::mgry_pow(s, instance(ctx, curve), a, e, out, n)); }
int ecsimd_hip_gfp_inverse(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out);
  if (curve != ECSIMD_HIP_P256 && curve != ECSIMD_HIP_SECP256K1) {
    // a run-time modulus.  Registered as PRIME: the shared division-step inversion below (the unique inverse = x^(p-2)); otherwise x^(p-2) bit by
    // bit, which is what gfp.h:42-44 computes whatever p is (for a composite p it is not an inverse, and the reference returns it all the same).
    gmod M; if (!lookup_modulus(curve, &M)) return bad(ctx, "unknown curve / field id");
    if ((M.flags & GMOD_PRIME) && !ctx->ref_square && !overlaps(out, a)) RUN(launch::gfield_inverse_batched(s, M, a, out, n));
    RUN(launch::gfield_unop(s, M, launch::F_INVERSE, a, out, n, ctx->ref_square != 0));
  }
  // one inversion per ~64 elements (Montgomery's trick, out[] as scratch) unless the call is in place
  // (the reference-square instances raise to p - 2 per element, squaring by squaring as the reference does)
  if (overlaps(out, a) || ctx->ref_square) RUN(launch::field_unop(s, instance(ctx, curve), launch::F_INVERSE, a, out, n));
  RUN(launch::inverse_batched(s, curve, a, out, n)); }
int ecsimd_hip_gfp_opposite(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, uint64_t* out, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out); FIELD_OR_CURVE(launch::gfield_unop(s, M, launch::F_OPPOSITE, a, out, n, false)); RUN(launch::field_unop(s, instance(ctx, curve), launch::F_OPPOSITE, a, out, n)); }
int ecsimd_hip_gfp_sqrt(ecsimd_hip_ctx* ctx, int curve, const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(a); REQUIRE_PTR(out);
  if (curve != ECSIMD_HIP_P256 && curve != ECSIMD_HIP_SECP256K1) {
    gmod M; if (!lookup_modulus(curve, &M)) return bad(ctx, "unknown curve / field id");
    if (!(M.flags & GMOD_3MOD4)) return bad(ctx, "gfp_sqrt needs p = 3 mod 4 (the reference's GFp does not instantiate otherwise: gfp.h:84)");
    RUN(launch::gfield_sqrt(s, M, a, out, ok, n, ctx->ref_square != 0));
  }
  RUN(launch::gfp_sqrt(s, instance(ctx, curve), a, out, ok, n)); }

// ---- L4/L5
int ecsimd_hip_from_affine(ecsimd_hip_ctx* ctx, int curve, const uint64_t* x, const uint64_t* y, uint64_t* jx, uint64_t* jy, uint64_t* jz, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(x); REQUIRE_PTR(y); REQUIRE_PTR(jx); REQUIRE_PTR(jy); REQUIRE_PTR(jz); GENERIC_CURVE(launch::gc_from_affine(s, GC, x, y, jx, jy, jz, n));
  REQUIRE_CURVE(); RUN(launch::from_affine(s, instance(ctx, curve), x, y, jx, jy, jz, n)); }
int ecsimd_hip_to_affine(ecsimd_hip_ctx* ctx, int curve, const uint64_t* jx, const uint64_t* jy, const uint64_t* jz, uint64_t* x, uint64_t* y, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(jx); REQUIRE_PTR(jy); REQUIRE_PTR(jz); REQUIRE_PTR(x);
  if (y && !aligned16(y)) return bad(ctx, "y is not 16-byte aligned");       // y == NULL: the x coordinate only
  // Simultaneous inversion uses x[] as scratch: only when the outputs do not alias the inputs.
  const bool alias = overlaps(x, jx) || overlaps(x, jy) || overlaps(x, jz) || (y && (overlaps(y, jx) || overlaps(y, jy) || overlaps(y, jz) || overlaps(x, y)));
  if (alias || ctx->ref_square) GENERIC_CURVE(launch::gc_to_affine(s, GC, jx, jy, jz, x, y, n, ctx->ref_square != 0));
  GENERIC_CURVE(launch::gc_to_affine_batched(s, GC, jx, jy, jz, x, y, n));
  REQUIRE_CURVE();
  if (alias || ctx->ref_square) RUN(launch::to_affine(s, instance(ctx, curve), jx, jy, jz, x, y, n));
  RUN(launch::to_affine_batched(s, curve, jx, jy, jz, x, y, n, false)); }
int ecsimd_hip_compute_y(ecsimd_hip_ctx* ctx, int curve, const uint64_t* x, uint64_t* y, uint8_t* ok, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(x); REQUIRE_PTR(y); GENERIC_CURVE(launch::gc_compute_y(s, GC, x, y, ok, n, ctx->ref_square != 0));
  REQUIRE_CURVE(); RUN(launch::compute_y(s, instance(ctx, curve), x, y, ok, n)); }
int ecsimd_hip_dblu(ecsimd_hip_ctx* ctx, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(px); REQUIRE_PTR(py); REQUIRE_PTR(pz); REQUIRE_PTR(rx); REQUIRE_PTR(ry); REQUIRE_PTR(rz); GENERIC_CURVE(launch::gc_dblu(s, GC, px, py, pz, rx, ry, rz, n, ctx->ref_square != 0));
  REQUIRE_CURVE(); RUN(launch::dblu(s, instance(ctx, curve), px, py, pz, rx, ry, rz, n)); }
int ecsimd_hip_zaddu(ecsimd_hip_ctx* ctx, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, const uint64_t* ox, const uint64_t* oy, const uint64_t* oz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(px); REQUIRE_PTR(py); REQUIRE_PTR(pz); REQUIRE_PTR(ox); REQUIRE_PTR(oy); REQUIRE_PTR(rx); REQUIRE_PTR(ry); REQUIRE_PTR(rz);
  (void)oz;   // co-Z: O.z == P.z by precondition (curve_group.h:92)
  GENERIC_CURVE(launch::gc_zaddu(s, GC, px, py, pz, ox, oy, rx, ry, rz, n, ctx->ref_square != 0));
  REQUIRE_CURVE(); RUN(launch::zaddu(s, instance(ctx, curve), px, py, pz, ox, oy, rx, ry, rz, n)); }
int ecsimd_hip_zdau(ecsimd_hip_ctx* ctx, int curve, const uint64_t* px, const uint64_t* py, const uint64_t* pz, uint64_t* qx, uint64_t* qy, uint64_t* qz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(px); REQUIRE_PTR(py); REQUIRE_PTR(pz); REQUIRE_PTR(qx); REQUIRE_PTR(qy); REQUIRE_PTR(qz); REQUIRE_PTR(rx); REQUIRE_PTR(ry); REQUIRE_PTR(rz);
  GENERIC_CURVE(launch::gc_zdau(s, GC, px, py, pz, qx, qy, qz, rx, ry, rz, n, ctx->ref_square != 0));
  REQUIRE_CURVE(); RUN(launch::zdau(s, instance(ctx, curve), px, py, pz, qx, qy, qz, rx, ry, rz, n)); }
int ecsimd_hip_zdau_repeat(ecsimd_hip_ctx* ctx, int curve, const uint64_t* px, const uint64_t* py, const uint64_t* pz, const uint64_t* qx, const uint64_t* qy,
                           uint64_t* rx, uint64_t* ry, uint64_t* sx, uint64_t* sy, uint64_t* oz, size_t n, int iters, uint64_t swap_bits, int radix) {
  REQUIRE_CTX(); REQUIRE_PTR(px); REQUIRE_PTR(py); REQUIRE_PTR(pz); REQUIRE_PTR(qx); REQUIRE_PTR(qy); REQUIRE_PTR(rx); REQUIRE_PTR(ry); REQUIRE_PTR(sx); REQUIRE_PTR(sy); REQUIRE_PTR(oz);
  if (iters < 0) return bad(ctx, "iters is negative");
  if (radix != 29 && radix != 32) return bad(ctx, "radix is 29 or 32");
  if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE && ctx->ref_square) return bad(ctx, "zdau_repeat on a registered curve has no reference-square form");
  GENERIC_CURVE(launch::gc_zdau_repeat(s, GC, px, py, pz, qx, qy, rx, ry, sx, sy, oz, n, iters, swap_bits, radix));
  REQUIRE_CURVE();
  if (radix == 29 && instance(ctx, curve) != curve) return bad(ctx, "the reduced-radix loop has no reference-square form (the dropped carry depends on the 32-bit Montgomery digits)");
  RUN(launch::zdau_repeat(s, instance(ctx, curve), px, py, pz, qx, qy, rx, ry, sx, sy, oz, n, iters, swap_bits, radix)); }
int ecsimd_hip_add_z2_1(ecsimd_hip_ctx* ctx, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(ax); REQUIRE_PTR(ay); REQUIRE_PTR(az); REQUIRE_PTR(bx); REQUIRE_PTR(by); REQUIRE_PTR(rx); REQUIRE_PTR(ry); REQUIRE_PTR(rz);
  GENERIC_CURVE(launch::gc_add_z2_1(s, GC, ax, ay, az, bx, by, rx, ry, rz, n, ctx->ref_square != 0));
  REQUIRE_CURVE(); RUN(launch::add_z2_1(s, instance(ctx, curve), ax, ay, az, bx, by, rx, ry, rz, n)); }
int ecsimd_hip_add_mixed_complete(ecsimd_hip_ctx* ctx, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* az, const uint64_t* bx, const uint64_t* by, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  REQUIRE_CTX(); REQUIRE_CURVE(); REQUIRE_PTR(ax); REQUIRE_PTR(ay); REQUIRE_PTR(az); REQUIRE_PTR(bx); REQUIRE_PTR(by); REQUIRE_PTR(rx); REQUIRE_PTR(ry); REQUIRE_PTR(rz);
  RUN(launch::add_mixed_complete(s, curve, ax, ay, az, bx, by, rx, ry, rz, n)); }
int ecsimd_hip_trplu(ecsimd_hip_ctx* ctx, int curve, uint64_t* px, uint64_t* py, uint64_t* pz, uint64_t* rx, uint64_t* ry, uint64_t* rz, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(px); REQUIRE_PTR(py); REQUIRE_PTR(pz); REQUIRE_PTR(rx); REQUIRE_PTR(ry); REQUIRE_PTR(rz); GENERIC_CURVE(launch::gc_trplu(s, GC, px, py, pz, rx, ry, rz, n, ctx->ref_square != 0));
  REQUIRE_CURVE(); RUN(launch::trplu(s, instance(ctx, curve), px, py, pz, rx, ry, rz, n)); }

int ecsimd_hip_scalar_mult(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  REQUIRE_CTX();
  // no base point AND the caller says so (BASE_GENERATOR): the generator.  Two null pointers alone are an error, as they were before round 4 -- a caller whose
  // point buffers failed to allocate must not get k G back (and ALG_WINDOWED means another algorithm on a fixed base than on a variable one).
  if ((flags & ECSIMD_HIP_BASE_GENERATOR) && x == nullptr && y == nullptr)
    return ecsimd_hip_scalar_mult_base(ctx, curve, k, ox, oy, oz, n, flags & ~(ECSIMD_HIP_BASE_MGRY | ECSIMD_HIP_BASE_GENERATOR));
  if (flags & ECSIMD_HIP_BASE_GENERATOR) return bad(ctx, "BASE_GENERATOR takes x = y = NULL");
  REQUIRE_PTR(k); REQUIRE_PTR(x); REQUIRE_PTR(y); REQUIRE_PTR(ox); REQUIRE_OUT_Y(oy);
  if (!(flags & ECSIMD_HIP_OUT_AFFINE)) REQUIRE_PTR(oz);
  if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) return run_gladder(ctx, curve, k, 4, x, y, ox, oy, oz, n, flags);
  REQUIRE_CURVE();
  if (flags & (ECSIMD_HIP_ALG_WINDOWED | ECSIMD_HIP_ALG_WINDOWED_SIGNED)) {
    // per-lane window tables (8 multiples of P) in HBM + signed 4-bit windows (k_varwin.inc): a different algorithm from
    // the reference ladder, so affine output only (SURVEY.md 8(a) level A)
    if (!(flags & ECSIMD_HIP_OUT_AFFINE)) return bad(ctx, "ALG_WINDOWED needs OUT_AFFINE");
    if ((flags & ECSIMD_HIP_ALG_CONSTANT_TIME) && (flags & ECSIMD_HIP_ALG_WINDOWED_SIGNED)) return bad(ctx, "ALG_CONSTANT_TIME modifies ALG_WINDOWED only, not ALG_WINDOWED_SIGNED");
    NO_COMPAT("ALG_WINDOWED");
    return run_varwin(ctx, curve, k, 4, x, y, ox, oy, n, flags, 0);
  }
  if (flags & ECSIMD_HIP_ALG_CONSTANT_TIME) return bad(ctx, "ALG_CONSTANT_TIME modifies ALG_WINDOWED (the ladder is constant-time as it is)");
  return run_ladder(ctx, curve, k, 4, x, y, ox, oy, oz, n, flags); }
// ---- host arrays in, host arrays out: the PCIe-inclusive form of scalar_mult.  Chunks of 2^19 elements alternate between this context and a helper context (a
// second HIP stream): while the ladder of one chunk runs, the calling thread copies the next chunk in on the other side and launches it, then waits for the
// previous chunk and copies it out -- the copies of one chunk overlap the ladder of its neighbour and the GPU always has a launch queued; the caller's arrays
// may be ordinary pageable memory.  Synchronous: returns with the results in place.
namespace {
constexpr size_t HOST_CHUNK = (size_t)1 << 19;
struct host_side { ecsimd_hip_ctx* c; uint64_t *k, *x, *y, *o[3]; };
int host_stage(ecsimd_hip_ctx* c, size_t elems, int arrays, host_side* S) {
  const size_t bytes = (size_t)arrays * elems * 32;
  if (c->hstage_bytes < bytes) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->hstage); c->hstage = nullptr; c->hstage_bytes = 0;
    hipError_t e = hipMalloc(&c->hstage, bytes);
    if (e != hipSuccess) return fail(c, e, "scalar_mult_host staging");
    c->hstage_bytes = bytes;
  }
  uint64_t* p = c->hstage;
  S->c = c; S->k = p; S->x = p + 4 * elems; S->y = p + 8 * elems;
  for (int j = 0; j < 3; ++j) S->o[j] = p + (size_t)(3 + j) * 4 * elems;
  return ECSIMD_HIP_OK;
}
}  // namespace
int ecsimd_hip_scalar_mult_host(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  REQUIRE_CTX();
  const bool generator = (flags & ECSIMD_HIP_BASE_GENERATOR) != 0, affine = (flags & ECSIMD_HIP_OUT_AFFINE) != 0;
  if (!k || !ox) return n ? bad(ctx, "k / ox is null") : ECSIMD_HIP_OK;
  if (generator ? (x || y) : (!x || !y)) return bad(ctx, "a base point array is missing (or BASE_GENERATOR came with one)");
  if (!affine && (!oy || !oz)) return bad(ctx, "a Jacobian result needs ox, oy and oz");
  if (n == 0) return ECSIMD_HIP_OK;
  if (capturing(ctx)) return bad(ctx, "scalar_mult_host copies and synchronises: not inside a stream capture");
  (void)hipSetDevice(ctx->device);
  const size_t chunk = n < HOST_CHUNK ? n : HOST_CHUNK, chunks = (n + chunk - 1) / chunk;
  if (chunks > 1 && !ctx->helper) { int rc = ecsimd_hip_init(ctx->device, &ctx->helper); if (rc != ECSIMD_HIP_OK) return bad(ctx, "scalar_mult_host: the helper context could not be created"); }
  host_side S[2];
  for (int i = 0; i < (chunks > 1 ? 2 : 1); ++i) {
    ecsimd_hip_ctx* c = i == 0 ? ctx : ctx->helper;
    c->ref_square = ctx->ref_square;
    int rc = host_stage(c, chunk, 6, &S[i]);
    if (rc != ECSIMD_HIP_OK) { if (c != ctx) snprintf(ctx->err, sizeof ctx->err, "%s", c->err); return rc; }
  }
  auto report = [&](ecsimd_hip_ctx* c, int rc) { if (c != ctx) snprintf(ctx->err, sizeof ctx->err, "%s", c->err); return rc; };
  auto launch = [&](size_t ci) -> int {
    host_side& s = S[ci & 1];
    const size_t first = ci * chunk, m = (n - first) < chunk ? (n - first) : chunk;
    int rc = ecsimd_hip_memcpy_h2d(s.c, s.k, k + 4 * first, m * 32);
    if (rc == ECSIMD_HIP_OK && !generator) rc = ecsimd_hip_memcpy_h2d(s.c, s.x, x + 4 * first, m * 32);
    if (rc == ECSIMD_HIP_OK && !generator) rc = ecsimd_hip_memcpy_h2d(s.c, s.y, y + 4 * first, m * 32);
    if (rc == ECSIMD_HIP_OK) rc = ecsimd_hip_scalar_mult(s.c, curve, s.k, generator ? nullptr : s.x, generator ? nullptr : s.y, s.o[0], oy ? s.o[1] : nullptr, affine ? nullptr : s.o[2], m, flags);
    return report(s.c, rc);
  };
  auto drain = [&](size_t ci) -> int {
    host_side& s = S[ci & 1];
    const size_t first = ci * chunk, m = (n - first) < chunk ? (n - first) : chunk;
    int rc = ecsimd_hip_memcpy_d2h(s.c, ox + 4 * first, s.o[0], m * 32);               // (waits for this side's kernels, then copies)
    if (rc == ECSIMD_HIP_OK && oy) rc = ecsimd_hip_memcpy_d2h(s.c, oy + 4 * first, s.o[1], m * 32);
    if (rc == ECSIMD_HIP_OK && !affine) rc = ecsimd_hip_memcpy_d2h(s.c, oz + 4 * first, s.o[2], m * 32);
    return report(s.c, rc);
  };
  for (size_t ci = 0; ci <= chunks; ++ci) {
    if (ci < chunks) { int rc = launch(ci); if (rc != ECSIMD_HIP_OK) { if (ci > 0) (void)ecsimd_hip_sync(S[(ci - 1) & 1].c); return rc; } }
    if (ci > 0) { int rc = drain(ci - 1); if (rc != ECSIMD_HIP_OK) return rc; }
  }
  return ECSIMD_HIP_OK;
}
int ecsimd_hip_scalar_mult_1s(ecsimd_hip_ctx* ctx, int curve, const uint64_t k1[4], const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  REQUIRE_CTX(); REQUIRE_PTR(x); REQUIRE_PTR(y); REQUIRE_PTR(ox); REQUIRE_OUT_Y(oy); if (!k1) return bad(ctx, "k1 is null");
  if (!(flags & ECSIMD_HIP_OUT_AFFINE)) REQUIRE_PTR(oz);
  if (curve < ECSIMD_HIP_FIRST_REGISTERED_CURVE) REQUIRE_CURVE();
  launch::words8 w; for (int i = 0; i < 4; ++i) { w.w[2 * i] = (uint32_t)k1[i]; w.w[2 * i + 1] = (uint32_t)(k1[i] >> 32); }
  uint32_t* kdev = ctx->sink + 1024 - 8;    // 32-byte aligned slot at the end of the scratch page
  if (n == 0) return ECSIMD_HIP_OK;
  (void)hipSetDevice(ctx->device);
  store_words(ctx->stream, w, kdev);
  if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) return run_gladder(ctx, curve, reinterpret_cast<const uint64_t*>(kdev), 0, x, y, ox, oy, oz, n, flags);
  if (flags & (ECSIMD_HIP_ALG_WINDOWED | ECSIMD_HIP_ALG_WINDOWED_SIGNED)) {
    if (!(flags & ECSIMD_HIP_OUT_AFFINE)) return bad(ctx, "ALG_WINDOWED needs OUT_AFFINE");
    if ((flags & ECSIMD_HIP_ALG_CONSTANT_TIME) && (flags & ECSIMD_HIP_ALG_WINDOWED_SIGNED)) return bad(ctx, "ALG_CONSTANT_TIME modifies ALG_WINDOWED only, not ALG_WINDOWED_SIGNED");
    NO_COMPAT("ALG_WINDOWED");
    return run_varwin(ctx, curve, reinterpret_cast<const uint64_t*>(kdev), 0, x, y, ox, oy, n, flags, 0);
  }
  if (flags & ECSIMD_HIP_ALG_CONSTANT_TIME) return bad(ctx, "ALG_CONSTANT_TIME modifies ALG_WINDOWED (the ladder is constant-time as it is)");
  return run_ladder(ctx, curve, reinterpret_cast<const uint64_t*>(kdev), 0, x, y, ox, oy, oz, n, flags); }
int ecsimd_hip_scalar_mult_base(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  REQUIRE_CTX(); REQUIRE_PTR(k); REQUIRE_PTR(ox); REQUIRE_OUT_Y(oy);
  flags &= ~ECSIMD_HIP_BASE_GENERATOR;                            // (implied here)
  if (!(flags & ECSIMD_HIP_OUT_AFFINE)) REQUIRE_PTR(oz);
  if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) {
    int rc = gc_small_base(ctx, curve, k, ox, oy, n, flags);       // small batches, affine output, no algorithm asked for: the constant-time comb, the ladder's bits kept
    if (rc != GC_NOT_TAKEN) return rc;
    return run_gladder(ctx, curve, k, 4, nullptr, nullptr, ox, oy, oz, n, flags & ~ECSIMD_HIP_BASE_MGRY);
  }
  REQUIRE_CURVE();
  if (flags & (ECSIMD_HIP_ALG_WINDOWED | ECSIMD_HIP_ALG_WINDOWED_SIGNED | ECSIMD_HIP_ALG_WINDOWED_BIG)) {
    const bool big = (flags & ECSIMD_HIP_ALG_WINDOWED_BIG) != 0;         // signed 20-bit windows, table in device memory
    const bool six = (flags & ECSIMD_HIP_ALG_WINDOWED_SIGNED) != 0;      // signed 7-bit windows, table in LDS
    // windows over a precomputed table, then one simultaneous inversion: affine output only
    // (the Jacobian representative differs from the reference ladder's -- SURVEY.md 8(a) level A).
    if (!(flags & ECSIMD_HIP_OUT_AFFINE)) return bad(ctx, "ALG_WINDOWED needs OUT_AFFINE");
    const bool ct = (flags & ECSIMD_HIP_ALG_CONSTANT_TIME) != 0;
    if (ct && !(ECS_FIXED4_ODD && ECS_SIGNED_ODD)) return bad(ctx, "this build (-DECS_FIXED4_ODD=0 / -DECS_SIGNED_ODD=0) has no constant-time comb");
    if (ct && (big || six)) return bad(ctx, "ALG_CONSTANT_TIME modifies ALG_WINDOWED only (its own 5-bit comb in LDS), not ALG_WINDOWED_SIGNED / ALG_WINDOWED_BIG");
    NO_COMPAT("ALG_WINDOWED");
    if (n == 0) return ECSIMD_HIP_OK;
    (void)hipSetDevice(ctx->device);
    const int ctbits = (ct && !six) ? (curve == ECSIMD_HIP_P256 ? CT_WBITS : CT_WBITS_SECP) : 0;   // the constant-time comb's own table (0: the 4-bit kernel)
    const bool ct6 = ctbits != 0;
    int rc = ensure_window_table(ctx, curve, big ? launch::BIG_WINDOW_BITS : six ? SIGNED_WBITS : ct6 ? ctbits : 4);
    if (rc == ECSIMD_HIP_OK) rc = ensure_workspace(ctx, 3 * n * 32);
    if (rc != ECSIMD_HIP_OK) return rc;
    uint64_t* jx = ctx->workspace; uint64_t* jy = jx + 4 * n; uint64_t* jz = jy + 4 * n;
    RUN(((big ? launch::base_windowed_big(s, curve, k, ctx->window16_table[curve], jx, jy, jz, n)
          : six ? launch::base_windowed_signed(s, curve, SIGNED_WBITS, k, ctx->window6_table[curve], jx, jy, jz, n, false)
          : ct6 ? launch::base_windowed_signed(s, curve, ctbits, k, ctx->windowct_table[curve], jx, jy, jz, n, true)
                : launch::base_windowed(s, curve, k, ctx->window_table[curve], jx, jy, jz, n, ct)),
         launch::to_affine_batched(s, curve, jx, jy, jz, ox, oy, n, true)));
  }
  if (flags & ECSIMD_HIP_ALG_CONSTANT_TIME) return bad(ctx, "ALG_CONSTANT_TIME modifies ALG_WINDOWED (the ladder is constant-time as it is)");
  // Small batches, affine output, no algorithm asked for: a launch of the ladder costs its 254 iterations however few lanes it has (1.3 ms), the
  // constant-time comb 51 additions (0.2 ms) -- and every reference test and benchmark multiplies G (benchs/curve_group.cpp:23-35).  The comb is as
  // safe for secret scalars as the ladder; its affine result is the true k*G, which is the ladder's everywhere but at the ladder's three degenerate
  // scalars, and there the lanes take the ladder's own (meaningless, but the reference's) coordinates from a 288-byte record: the same bits out.
  if ((flags & ECSIMD_HIP_OUT_AFFINE) && n <= SMALL_BASE_MAX && !(flags & (ECSIMD_HIP_REF_SQUARE_COMPAT | ECSIMD_HIP_LADDER_RADIX32)) && !ctx->ref_square &&
      ECS_FIXED4_ODD && ECS_SIGNED_ODD && !capturing_needs_build(ctx, curve)) {
    if (n == 0) return ECSIMD_HIP_OK;
    (void)hipSetDevice(ctx->device);
    const int ctbits = curve == ECSIMD_HIP_P256 ? CT_WBITS : CT_WBITS_SECP;
    if (ctbits) {
      int rc = ensure_window_table(ctx, curve, ctbits);
      const bool x_is_exact = (oy == nullptr && curve == ECSIMD_HIP_P256);    // the P-256 x-only ladder returns the true x for every k already
      if (rc == ECSIMD_HIP_OK && !x_is_exact) rc = ensure_base_special(ctx, curve);
      if (rc == ECSIMD_HIP_OK) rc = ensure_workspace(ctx, 3 * n * 32);
      if (rc != ECSIMD_HIP_OK) return rc;
      uint64_t* jx = ctx->workspace; uint64_t* jy = jx + 4 * n; uint64_t* jz = jy + 4 * n;
      hipStream_t s = ctx->stream;
      launch::base_windowed_signed(s, curve, ctbits, k, ctx->windowct_table[curve], jx, jy, jz, n, true);
      launch::to_affine_batched(s, curve, jx, jy, jz, ox, oy, n, true);
      if (!x_is_exact) launch::patch_special(s, k, ctx->base_special[curve], ox, oy, n);
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "scalar_mult_base (small batch) launch");
    }
  }
  return run_ladder(ctx, curve, k, 4, nullptr, nullptr, ox, oy, oz, n, flags); }
int ecsimd_hip_affine_add(ecsimd_hip_ctx* ctx, int curve, const uint64_t* ax, const uint64_t* ay, const uint64_t* bx, const uint64_t* by,
                          uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n) {
  REQUIRE_CTX(); if (curve < ECSIMD_HIP_FIRST_REGISTERED_CURVE) REQUIRE_CURVE(); REQUIRE_PTR(ax); REQUIRE_PTR(ay); REQUIRE_PTR(bx); REQUIRE_PTR(by); REQUIRE_PTR(rx);
  if (ry && !aligned16(ry)) return bad(ctx, "ry is not 16-byte aligned");
  if (overlaps(rx, ax) || overlaps(rx, ay) || overlaps(rx, bx) || overlaps(rx, by)) return bad(ctx, "rx must not alias an input (it is the inversion scratch)");
  GENERIC_CURVE(launch::gc_affine_add_batched(s, GC, ax, ay, bx, by, rx, ry, finite, n));
  RUN(launch::affine_add_batched(s, curve, ax, ay, bx, by, rx, ry, finite, n)); }

// u1[i]*G + u2[i]*Q[i]: windowed fixed-base product + windowed variable-base product + one batched affine
// addition, in chunks of VARWIN_CHUNK elements.  Every Q[i] is validated first (x, y < p, on the curve; the point at
// infinity (0, 0) fails): an invalid public key yields the point at infinity and finite[i] = 0 -- without the check
// the window tables of such a lane would hold multiples on some other curve and the result would look like a point.
// u1*G comes from the 20-bit window table in device memory (436 MB per curve, built on first use) once that table
// exists or the batch is large enough to pay for building it; smaller batches take the signed 7-bit table in LDS.
namespace {
constexpr size_t BIG_TABLE_WORTH_IT = (size_t)1 << 16;
int double_scalar_mult_impl(ecsimd_hip_ctx* ctx, int curve, const uint64_t* u1, const uint64_t* u2, const uint64_t* qx, const uint64_t* qy,
                            uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n, size_t reserve_behind) {
  (void)hipSetDevice(ctx->device);
  const size_t chunk = n < VARWIN_CHUNK ? n : VARWIN_CHUNK;
  const bool big = ctx->window16_table[curve] != nullptr || n >= BIG_TABLE_WORTH_IT;
  int rc = ensure_window_table(ctx, curve, big ? launch::BIG_WINDOW_BITS : SIGNED_WBITS);
  if (rc == ECSIMD_HIP_OK) rc = ensure_workspace(ctx, 7 * chunk * 32 + launch::varwin_scratch_bytes(chunk) + reserve_behind);   // 3 Jacobian + 2 x 2 affine + tables
  if (rc == ECSIMD_HIP_OK) rc = ensure_valid(ctx, (n + 15) / 16 * 16);
  if (rc != ECSIMD_HIP_OK) return rc;
  uint64_t* jx = ctx->workspace; uint64_t* jy = jx + 4 * chunk; uint64_t* jz = jy + 4 * chunk;
  uint64_t* gx = jz + 4 * chunk; uint64_t* gy = gx + 4 * chunk; uint64_t* px = gy + 4 * chunk; uint64_t* py = px + 4 * chunk;
  uint64_t* scratch = py + 4 * chunk;
  hipStream_t s = ctx->stream;
  launch::on_curve(s, curve, qx, qy, ctx->valid, n);
  for (size_t first = 0; first < n; first += chunk) {
    const size_t m = (n - first) < chunk ? (n - first) : chunk;
    if (big) launch::base_windowed_big(s, curve, u1 + 4 * first, ctx->window16_table[curve], jx, jy, jz, m);   // u1*G
    else launch::base_windowed_signed(s, curve, SIGNED_WBITS, u1 + 4 * first, ctx->window6_table[curve], jx, jy, jz, m, false);
    launch::to_affine_batched(s, curve, jx, jy, jz, gx, gy, m, true);
    launch::varwin_scalar_mult(s, curve, u2 + 4 * first, 4, qx + 4 * first, qy + 4 * first, ECSIMD_HIP_BASE_CLASSICAL, scratch, px, py, m);   // u2*Q
    launch::affine_add_batched(s, curve, gx, gy, px, py, rx + 4 * first, ry ? ry + 4 * first : nullptr, finite ? finite + first : nullptr, m);
  }
  launch::clear_invalid(s, ctx->valid, rx, ry, finite, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "double_scalar_mult launch");
}
}  // namespace

// ---- u1 G + u2 Q, ECDSA verification and signing on a curve registered at run time: with the group order (n >= 2^255) u1 G from the generator's comb
// (k_gcomb.hip) and u2 Q from the lane's own window table (k_gvarwin.hip: verification's scalars are public); without it the reference's ladder twice, the
// scalars kept clear of its three degenerate values (k_gc_ladder_safe_scalars); one shared inversion per product, a batched affine addition.
namespace {
constexpr size_t GC_CHUNK = (size_t)1 << 22;
constexpr size_t GC_BIG_TABLE_WORTH_IT = (size_t)1 << 20;          // u1 G + u2 Q on a registered curve: batches from here on build the 20-bit comb (0.3 s, 436 MB) on first use
struct gc_layout { size_t chunk; uint64_t *adj1, *adj2, *j[3], *gx, *gy, *px, *py, *win; uint8_t *neg1, *neg2; size_t bytes; };
// win: u2 Q goes through the window loop (k_gvarwin.hip), whose per-lane tables follow the nine arrays
gc_layout gc_plan(uint64_t* base, size_t n, bool win = false) {
  gc_layout L; L.chunk = n < GC_CHUNK ? n : GC_CHUNK;
  uint64_t* p = base; const size_t e = 4 * L.chunk;
  L.adj1 = p; p += e; L.adj2 = p; p += e; for (int i = 0; i < 3; ++i) { L.j[i] = p; p += e; }
  L.gx = p; p += e; L.gy = p; p += e; L.px = p; p += e; L.py = p; p += e;
  const size_t wbytes = win ? launch::gc_varwin_scratch_bytes(L.chunk) : 0;
  L.win = win ? p : nullptr; p += wbytes / 8;
  L.neg1 = reinterpret_cast<uint8_t*>(p); L.neg2 = L.neg1 + ((L.chunk + 15) / 16) * 16;
  L.bytes = 9 * L.chunk * 32 + wbytes + 2 * (((L.chunk + 15) / 16) * 16);
  return L;
}
// The comb of a registered curve (k_gcomb.hip): 64 windows x 8 odd multiples (2d + 1) 16^w G, then k* G and the record {k*, 0} (k_affine.inc comb_special's
// layout).  Needs the order (the recoding works modulo n) with n >= 2^255 (k mod n by one subtraction).  Every entry comes from the reference's ladder on this
// curve -- whose degenerate scalars the entries' multipliers must not be (checked; k* by way of n - k* if it is one) -- through the shared inversion.
bool gc_comb_possible(const curve_record& rec) { return rec.has_order && (rec.n.l[3] >> 63) != 0; }
bool gc_window_possible(const curve_record& rec) { return gc_comb_possible(rec) && rec.prime_order; }    // a variable base: every point has order n (curve_record)
// bits = 4: that table (summed from the top); bits = 7 / 5 / 20: the signed comb's 37 windows x 64 / the constant-time comb's 52 windows x 16 / the device-memory
// comb's 13 windows x 2^19 (436 MB; 1.3 GB of temporary memory and 6.8 M ladder passes, ~0.3 s, at its build) odd multiples (2d + 1) 2^(bits w) G (summed from
// the bottom; the top window's digit is at most 15 / 1 / 65 535: its other entries are never read and hold G).
int ensure_gc_comb(ecsimd_hip_ctx* ctx, int curve, const curve_record& rec, const uint32_t** out, int bits = 4) {
  auto slot = [bits](ecsimd_hip_ctx::gcomb_entry& t) -> uint32_t*& { return bits == 4 ? t.table : bits == 7 ? t.table7 : bits == 20 ? t.table20 : t.table5; };
  for (auto& t : ctx->gcomb) if (t.curve == curve && slot(t)) { *out = slot(t); return ECSIMD_HIP_OK; }
  if (!gc_comb_possible(rec)) return bad(ctx, "the windowed algorithms on a registered curve need its group order n, n >= 2^255");
  if (capturing(ctx)) return bad(ctx, "a window table would have to be built during stream capture: run this call once before capturing");
  const int W = bits == 4 ? launch::GCOMB_WINDOWS : bits == 7 ? launch::GCOMB7_WINDOWS : bits == 20 ? launch::GCOMB20_WINDOWS : launch::GCOMB5_WINDOWS;
  const int PER = bits == 4 ? launch::GCOMB_ENTRIES : bits == 7 ? launch::GCOMB7_ENTRIES : bits == 20 ? launch::GCOMB20_ENTRIES : launch::GCOMB5_ENTRIES;
  const size_t table_entries = (size_t)W * PER, entries = table_entries + 1;
  std::vector<uint64_t> host_k;
  try { host_k.assign(entries * 4, 0); ctx->gcomb.reserve(ctx->gcomb.size() + 1); } catch (...) { return bad(ctx, "window table: out of host memory"); }
  for (int w = 0; w < W; ++w)
    for (int d = 0; d < PER; ++d) {
      uint64_t* e = &host_k[((size_t)w * PER + d) * 4];
      const int pos = bits * w, limb = pos / 64, off = pos % 64;
      const unsigned __int128 v = (unsigned __int128)(2u * (unsigned)d + 1u) << off;
      if (limb + 1 >= 4 && (uint64_t)(v >> 64) != 0) { e[0] = 1; continue; }        // (2d + 1) 2^pos >= 2^256: beyond the top digit's range, never read
      e[limb] = (uint64_t)v;
      if (limb + 1 < 4) e[limb + 1] = (uint64_t)(v >> 64);
      u256 m; for (int l = 0; l < 4; ++l) m.l[l] = e[l];
      if (u_ladder_degenerate(rec.n, m)) return bad(ctx, "window table: a table multiplier is one of the ladder's degenerate scalars on this curve");
    }
  // 4 bits, summed from the top: k* = n - 2 (n mod 16), and only if bit 4 of it is clear (k_affine.inc comb_special); 7 / 5 bits, summed from the bottom:
  // k* = n - 2 (n mod 2^(bits (W - 1))) (5 bits: 2^256 - n, one of the ladder's degenerate scalars).  Its point by way of n - k* if the ladder cannot do k*
  uint64_t kstar[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool negate_special = false;
  {
    u256 m2, ks;
    if (bits == 4) m2 = u256{{2 * (rec.n.l[0] & 15u), 0, 0, 0}};
    else {
      const int low = bits * (W - 1);                                            // 252 / 255
      u256 m = rec.n;
      for (int l = 0; l < 4; ++l) { const int lo = 64 * l; if (low <= lo) m.l[l] = 0; else if (low < lo + 64) m.l[l] &= (1ull << (low - lo)) - 1ull; }
      for (int l = 3; l > 0; --l) m2.l[l] = (m.l[l] << 1) | (m.l[l - 1] >> 63);
      m2.l[0] = m.l[0] << 1;                                                     // 2 m < n (m = n mod 2^252 < 2^252; m = n - 2^255 < n / 2)
    }
    (void)u_sub(ks, rec.n, m2);
    const bool have = bits == 4 ? ((ks.l[0] >> 4) & 1u) == 0 : !u_is_zero(ks);
    if (have) for (int l = 0; l < 4; ++l) kstar[l] = ks.l[l];
    u256 mult = have ? ks : u256{{1, 0, 0, 0}};                  // without a k*: any scalar, the point is never used
    if (have && u_ladder_degenerate(rec.n, ks)) {
      u256 alt; (void)u_sub(alt, rec.n, ks);
      if (u_is_zero(alt) || u_ladder_degenerate(rec.n, alt)) return bad(ctx, "window table: neither k* nor n - k* is a scalar the ladder multiplies correctly");
      mult = alt; negate_special = true;
    }
    for (int l = 0; l < 4; ++l) host_k[table_entries * 4 + l] = mult.l[l];
  }
  (void)hipSetDevice(ctx->device);
  uint64_t* kd = nullptr;
  uint32_t* table = nullptr;
  hipError_t e = hipMalloc(&kd, 6 * entries * 32 + 64);
  if (e == hipSuccess) e = hipMalloc(&table, (entries + 1) * 64);
  uint64_t* tx = kd + entries * 4; uint64_t* ty = tx + entries * 4;
  uint64_t* jx = ty + entries * 4; uint64_t* jy = jx + entries * 4; uint64_t* jz = jy + entries * 4;
  uint8_t* flag = reinterpret_cast<uint8_t*>(jz + entries * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(kd, host_k.data(), entries * 32, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(flag, 1, 16, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // host_k must outlive the copy
  if (e == hipSuccess) {
    launch::gc_scalar_mult(ctx->stream, rec.G, kd, 4, nullptr, nullptr, jx, jy, jz, entries, 0);
    launch::gc_to_affine_batched(ctx->stream, rec.G, jx, jy, jz, tx, ty, entries);
    if (negate_special) launch::gc_negate_where(ctx->stream, rec.G, flag, ty + table_entries * 4, 1);
    launch::gc_pack_table(ctx->stream, rec.G, tx, ty, table, (int)entries);
    e = hipMemcpyAsync(table + entries * 16, kstar, 64, hipMemcpyHostToDevice, ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipGetLastError();
  (void)hipFree(kd);
  if (e != hipSuccess) { (void)hipFree(table); return fail(ctx, e, "window table build (registered curve)"); }
  bool placed = false;
  for (auto& t : ctx->gcomb) if (t.curve == curve) { slot(t) = table; placed = true; }
  if (!placed) { ctx->gcomb.push_back({curve, nullptr, nullptr, nullptr, nullptr, nullptr}); slot(ctx->gcomb.back()) = table; }
  *out = table;
  return ECSIMD_HIP_OK;
}
// The small-batch route's record for a registered curve (ensure_base_special for the built-in ones): n - 1, 2^256 - n - 1, 2^256 - n and what the reference's
// ladder returns for them on G, affine -- computed by the very launches run_gladder(OUT_AFFINE) makes.
int ensure_gc_base_special(ecsimd_hip_ctx* ctx, int curve, const curve_record& rec, const uint64_t** out) {
  for (auto& t : ctx->gcomb) if (t.curve == curve && t.special) { *out = t.special; return ECSIMD_HIP_OK; }
  if (capturing(ctx)) return bad(ctx, "a record would have to be built during stream capture: run this call once before capturing");
  u256 one = {{1, 0, 0, 0}}, zero = {{0, 0, 0, 0}}, s3[3];
  (void)u_sub(s3[0], rec.n, one); (void)u_sub(s3[2], zero, rec.n); (void)u_sub(s3[1], s3[2], one);
  uint64_t host[12];
  for (int j = 0; j < 3; ++j) { if (!u_ladder_degenerate(rec.n, s3[j])) return bad(ctx, "base_special: not a degenerate scalar"); for (int l = 0; l < 4; ++l) host[4 * j + l] = s3[j].l[l]; }
  uint64_t* r = nullptr; uint64_t* tmp = nullptr;
  hipError_t e = hipMalloc(&r, 9 * 32);
  if (e == hipSuccess) e = hipMalloc(&tmp, 9 * 32);
  if (e == hipSuccess) e = hipMemcpyAsync(r, host, 3 * 32, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) {
    launch::gc_scalar_mult(ctx->stream, rec.G, r, 4, nullptr, nullptr, tmp, tmp + 12, tmp + 24, 3, 0);
    launch::gc_to_affine_batched(ctx->stream, rec.G, tmp, tmp + 12, tmp + 24, r + 12, r + 24, 3);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  (void)hipFree(tmp);
  if (e != hipSuccess) { (void)hipFree(r); return fail(ctx, e, "base_special build (registered curve)"); }
  for (auto& t : ctx->gcomb) if (t.curve == curve) { t.special = r; *out = r; return ECSIMD_HIP_OK; }
  (void)hipFree(r);
  return bad(ctx, "base_special: the curve's table is missing");
}
launch::words8 order_words(const curve_record& rec) { launch::words8 w; for (int i = 0; i < 4; ++i) { w.w[2 * i] = (uint32_t)rec.n.l[i]; w.w[2 * i + 1] = (uint32_t)(rec.n.l[i] >> 32); } return w; }
}  // namespace
namespace {
// scalar_mult_base(registered curve, ALG_WINDOWED [| ALG_CONSTANT_TIME] | OUT_AFFINE): the comb, then the shared inversion -- the true k G for every k
// (k mod n = 0: (0, 0)), which is the ladder's affine result everywhere but at the ladder's degenerate scalars.
int run_gcomb(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, uint64_t* ox, uint64_t* oy, size_t n, int flags) {
  if (!(flags & ECSIMD_HIP_OUT_AFFINE)) return bad(ctx, "ALG_WINDOWED needs OUT_AFFINE");
  if (ctx->ref_square || (flags & (ECSIMD_HIP_REF_SQUARE_COMPAT | ECSIMD_HIP_LADDER_RADIX32))) return bad(ctx, "ALG_WINDOWED is not the reference's algorithm: no ECSIMD_HIP_REF_SQUARE_COMPAT / LADDER_RADIX32 form");
  curve_record rec; if (!lookup_curve_record(curve, &rec)) return bad(ctx, "unknown curve id");
  if (n == 0) return gc_comb_possible(rec) ? ECSIMD_HIP_OK : bad(ctx, "the windowed algorithms on a registered curve need its group order n, n >= 2^255");
  if (n > (size_t)0x7fffffff * BLOCK) return bad(ctx, "batch too large");
  (void)hipSetDevice(ctx->device);
  const uint32_t* table = nullptr;
  const bool seven = (flags & ECSIMD_HIP_ALG_WINDOWED_SIGNED) != 0;     // signed 7-bit windows in 148 KiB of LDS: 36 additions instead of 63 (public scalars)
  const bool big = (flags & ECSIMD_HIP_ALG_WINDOWED_BIG) != 0;          // 20-bit windows over 436 MB in device memory: 12 additions (public scalars)
  const bool five = !seven && !big && (flags & ECSIMD_HIP_ALG_CONSTANT_TIME) != 0;   // the constant-time comb: 5-bit windows, 51 additions, every entry of a window read
  int rc = ensure_gc_comb(ctx, curve, rec, &table, seven ? 7 : big ? 20 : five ? 5 : 4);
  if (rc == ECSIMD_HIP_OK) rc = ensure_workspace(ctx, 3 * n * 32);
  if (rc != ECSIMD_HIP_OK) return rc;
  uint64_t* jx = ctx->workspace; uint64_t* jy = jx + 4 * n; uint64_t* jz = jy + 4 * n;
  if (seven || five || big) launch::gc_base_windowed_s(ctx->stream, rec.G, order_words(rec), seven ? 7 : big ? 20 : 5, k, table, jx, jy, jz, n);
  else launch::gc_base_windowed(ctx->stream, rec.G, order_words(rec), k, table, jx, jy, jz, n, false);
  launch::gc_to_affine_batched(ctx->stream, rec.G, jx, jy, jz, ox, oy, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "scalar_mult_base (registered curve, windowed) launch");
}
// scalar_mult(registered curve, ALG_WINDOWED [| ALG_CONSTANT_TIME] | OUT_AFFINE) on a variable base: per-lane tables of the eight odd multiples of P over one Z, the
// window loop on the isomorphic curve (k_gvarwin.hip), then the shared inversion -- the true k P for every k (k = 0 mod n: (0, 0)).  In chunks of 2^22 lanes
// (640 B of scratch per lane).  `reserve` bytes at the start of the workspace stay untouched (gc_double_scalar_mult).
int run_gvarwin(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, int k_stride, const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy, size_t n, int flags, size_t reserve) {
  if (!(flags & ECSIMD_HIP_OUT_AFFINE)) return bad(ctx, "ALG_WINDOWED needs OUT_AFFINE");
  if (ctx->ref_square || (flags & ECSIMD_HIP_REF_SQUARE_COMPAT)) return bad(ctx, "ALG_WINDOWED is not the reference's algorithm: no ECSIMD_HIP_REF_SQUARE_COMPAT form");
  if (flags & ECSIMD_HIP_LADDER_RADIX32) return bad(ctx, "LADDER_RADIX32 selects a ladder loop: not with ALG_WINDOWED");
  curve_record rec; if (!lookup_curve_record(curve, &rec)) return bad(ctx, "unknown curve id");
  if (!gc_comb_possible(rec)) return bad(ctx, "the windowed algorithms on a registered curve need its group order n, n >= 2^255");
  if (!gc_window_possible(rec)) return bad(ctx, "ALG_WINDOWED on a variable base needs a group of prime order: n is not a prime in p's Hasse interval (the ladder has no such condition)");
  if (n == 0) return ECSIMD_HIP_OK;
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return fail(ctx, e, "hipSetDevice");
  const size_t chunk = n < VARWIN_CHUNK ? n : VARWIN_CHUNK;
  int rc = ensure_workspace(ctx, reserve + launch::gc_varwin_scratch_bytes(chunk));
  if (rc != ECSIMD_HIP_OK) return rc;
  uint64_t* scratch = ctx->workspace + reserve / 8;
  for (size_t first = 0; first < n; first += chunk) {
    const size_t m = (n - first) < chunk ? (n - first) : chunk;
    launch::gc_varwin_scalar_mult(ctx->stream, rec.G, order_words(rec), k + (size_t)k_stride * first, k_stride, x + 4 * first, y + 4 * first, flags & (ECSIMD_HIP_BASE_MGRY | ECSIMD_HIP_ALG_CONSTANT_TIME),
                                  scratch, ox + 4 * first, oy ? oy + 4 * first : nullptr, m);
  }
  e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "scalar_mult (registered curve, windowed) launch");
}
// scalar_mult_base(registered curve, OUT_AFFINE, no ALG_* / ladder flag) on up to 2^16 lanes: as for the built-in curves (ecsimd_hip_scalar_mult_base below) a
// ladder launch costs its 254 iterations however few lanes it has; the constant-time comb costs 63 additions.  Its affine result is the true k G -- the ladder's
// everywhere but at the ladder's three degenerate scalars, and there the lanes take the ladder's own coordinates from the record: the same bits out.
int gc_small_base(ecsimd_hip_ctx* ctx, int curve, const uint64_t* k, uint64_t* ox, uint64_t* oy, size_t n, int flags) {
  constexpr int other = ECSIMD_HIP_REF_SQUARE_COMPAT | ECSIMD_HIP_LADDER_RADIX32 | ECSIMD_HIP_ALG_WINDOWED | ECSIMD_HIP_ALG_WINDOWED_SIGNED | ECSIMD_HIP_ALG_WINDOWED_BIG |
                        ECSIMD_HIP_ALG_NO_ENDOMORPHISM | ECSIMD_HIP_ALG_CONSTANT_TIME;
  if (!(flags & ECSIMD_HIP_OUT_AFFINE) || (flags & other) || ctx->ref_square || n == 0 || n > SMALL_BASE_MAX) return GC_NOT_TAKEN;
  curve_record rec; if (!lookup_curve_record(curve, &rec) || !gc_comb_possible(rec)) return GC_NOT_TAKEN;
  (void)hipSetDevice(ctx->device);
  const uint32_t* table = nullptr; const uint64_t* special = nullptr;
  int rc = ensure_gc_comb(ctx, curve, rec, &table);
  if (rc == ECSIMD_HIP_OK) rc = ensure_gc_base_special(ctx, curve, rec, &special);
  if (rc != ECSIMD_HIP_OK) { ctx->err[0] = 0; return GC_NOT_TAKEN; }                     // (a capture in progress and nothing built yet, a curve whose table cannot be built: the ladder)
  if (capturing(ctx) && ctx->workspace_bytes < 3 * n * 32) return GC_NOT_TAKEN;
  rc = ensure_workspace(ctx, 3 * n * 32);
  if (rc != ECSIMD_HIP_OK) return rc;
  uint64_t* jx = ctx->workspace; uint64_t* jy = jx + 4 * n; uint64_t* jz = jy + 4 * n;
  hipStream_t s = ctx->stream;
  launch::gc_base_windowed(s, rec.G, order_words(rec), k, table, jx, jy, jz, n, true);
  launch::gc_to_affine_batched(s, rec.G, jx, jy, jz, ox, oy, n);
  launch::patch_special(s, k, special, ox, oy, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "scalar_mult_base (registered curve, small batch) launch");
}
int gc_require_ecdsa(ecsimd_hip_ctx* ctx, int curve, curve_record* rec) {
  if (!lookup_curve_record(curve, rec)) return bad(ctx, "unknown curve id");
  if (!rec->has_order) return bad(ctx, "this curve was registered without its group order n");
  if (!rec->ecdsa_ok) return bad(ctx, "ECDSA on a registered curve needs p < 2n and n - u a good ladder scalar for the ladder's three degenerate u");
  if (ctx->ref_square) return bad(ctx, "not the reference's algorithm: no ECSIMD_HIP_REF_SQUARE_COMPAT form");
  return ECSIMD_HIP_OK;
}
// one product k P (P = G when x == nullptr) -> affine classical (ox, oy), correct for EVERY k < n; oy may be null (x only: the negation does not touch x)
void gc_safe_mult(hipStream_t s, const curve_record& rec, const gc_layout& L, uint64_t* adj, uint8_t* neg, const uint64_t* k, const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy, size_t m) {
  launch::gc_ladder_safe_scalars(s, rec.N, k, adj, neg, m);
  launch::gc_scalar_mult(s, rec.G, adj, 4, x, y, L.j[0], L.j[1], L.j[2], m, 0);
  launch::gc_to_affine_batched(s, rec.G, L.j[0], L.j[1], L.j[2], ox, oy, m);
  if (oy) launch::gc_negate_where(s, rec.G, neg, oy, m);
}
int gc_double_scalar_mult(ecsimd_hip_ctx* ctx, int curve_of, const curve_record& rec, const uint64_t* u1, const uint64_t* u2, const uint64_t* qx, const uint64_t* qy,
                          uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n, size_t reserve_behind) {
  (void)hipSetDevice(ctx->device);
  const bool win = gc_window_possible(rec);                         // u2 Q from the lane's own window table where the curve has a prime order n >= 2^255, else a ladder pass
  gc_layout L = gc_plan(nullptr, n, win);
  int rc = ensure_workspace(ctx, L.bytes + reserve_behind);
  if (rc == ECSIMD_HIP_OK) rc = ensure_valid(ctx, (n + 15) / 16 * 16);
  if (rc != ECSIMD_HIP_OK) return rc;
  const uint32_t* comb = nullptr;                                   // u1 G from the generator's table where the curve has one (n >= 2^255), else a ladder pass
  const uint32_t* comb7 = nullptr;                                  // ... preferably the signed 7-bit comb (u1 is public: 36 additions instead of 63)
  const uint32_t* comb20 = nullptr;                                 // ... or the 20-bit comb in device memory (12 additions) once it exists or the batch pays for its 436 MB
  if (gc_comb_possible(rec)) {
    bool have20 = false;
    for (auto& t : ctx->gcomb) if (t.curve == curve_of && t.table20) have20 = true;
    if (have20 || (n >= GC_BIG_TABLE_WORTH_IT && !capturing(ctx))) { rc = ensure_gc_comb(ctx, curve_of, rec, &comb20, 20); if (rc != ECSIMD_HIP_OK) comb20 = nullptr; }
    rc = comb20 ? ECSIMD_HIP_OK : ensure_gc_comb(ctx, curve_of, rec, &comb7, 7);
    if (rc != ECSIMD_HIP_OK) {
      if (!capturing(ctx)) return rc;
      comb7 = nullptr;
      rc = ensure_gc_comb(ctx, curve_of, rec, &comb); if (rc != ECSIMD_HIP_OK) comb = nullptr;      // (no table yet and a capture in progress: the other table, else the ladder)
    }
  }
  L = gc_plan(ctx->workspace, n, win);
  hipStream_t s = ctx->stream;
  launch::gc_on_curve(s, rec.G, qx, qy, ctx->valid, n);
  for (size_t first = 0; first < n; first += L.chunk) {
    const size_t m = (n - first) < L.chunk ? (n - first) : L.chunk;
    if (comb20 || comb7 || comb) {
      if (comb20) launch::gc_base_windowed_s(s, rec.G, order_words(rec), 20, u1 + 4 * first, comb20, L.j[0], L.j[1], L.j[2], m);
      else if (comb7) launch::gc_base_windowed_s(s, rec.G, order_words(rec), 7, u1 + 4 * first, comb7, L.j[0], L.j[1], L.j[2], m);
      else launch::gc_base_windowed(s, rec.G, order_words(rec), u1 + 4 * first, comb, L.j[0], L.j[1], L.j[2], m, false);
      launch::gc_to_affine_batched(s, rec.G, L.j[0], L.j[1], L.j[2], L.gx, L.gy, m);
    } else gc_safe_mult(s, rec, L, L.adj1, L.neg1, u1 + 4 * first, nullptr, nullptr, L.gx, L.gy, m);                       // u1 G
    if (win) launch::gc_varwin_scalar_mult(s, rec.G, order_words(rec), u2 + 4 * first, 4, qx + 4 * first, qy + 4 * first, 0, L.win, L.px, L.py, m);   // u2 Q (public scalars)
    else gc_safe_mult(s, rec, L, L.adj2, L.neg2, u2 + 4 * first, qx + 4 * first, qy + 4 * first, L.px, L.py, m);
    launch::gc_affine_add_batched(s, rec.G, L.gx, L.gy, L.px, L.py, rx + 4 * first, ry ? ry + 4 * first : nullptr, finite ? finite + first : nullptr, m);
  }
  launch::clear_invalid(s, ctx->valid, rx, ry, finite, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "double_scalar_mult (registered curve) launch");
}
int gc_ecdsa_verify_rx(ecsimd_hip_ctx* ctx, int curve_of, const curve_record& rec, const uint64_t* u1, const uint64_t* u2, const uint64_t* qx, const uint64_t* qy, const uint64_t* r, uint8_t* ok, size_t n, size_t extra) {
  const gc_layout L0 = gc_plan(nullptr, n, gc_window_possible(rec));
  const size_t behind = n * 32 + ((n + 15) / 16) * 16;
  int rc = ensure_workspace(ctx, L0.bytes + behind + extra);
  if (rc != ECSIMD_HIP_OK) return rc;
  uint64_t* rx = ctx->workspace + L0.bytes / 8;
  uint8_t* fin = reinterpret_cast<uint8_t*>(rx + 4 * n);
  rc = gc_double_scalar_mult(ctx, curve_of, rec, u1, u2, qx, qy, rx, nullptr, fin, n, behind + extra);
  if (rc != ECSIMD_HIP_OK) return rc;
  launch::gc_x_mod_n_equals(ctx->stream, rec.N, rx, fin, r, ok, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "ecdsa_verify_rx (registered curve) launch");
}
}  // namespace

int ecsimd_hip_on_curve(ecsimd_hip_ctx* ctx, int curve, const uint64_t* x, const uint64_t* y, uint8_t* ok, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(x); REQUIRE_PTR(y); if (!ok && n) return bad(ctx, "ok is null");
  GENERIC_CURVE(launch::gc_on_curve(s, GC, x, y, ok, n));
  REQUIRE_CURVE(); RUN(launch::on_curve(s, curve, x, y, ok, n)); }

int ecsimd_hip_double_scalar_mult(ecsimd_hip_ctx* ctx, int curve, const uint64_t* u1, const uint64_t* u2, const uint64_t* qx, const uint64_t* qy,
                                  uint64_t* rx, uint64_t* ry, uint8_t* finite, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(u1); REQUIRE_PTR(u2); REQUIRE_PTR(qx); REQUIRE_PTR(qy); REQUIRE_PTR(rx);
  if (ry && !aligned16(ry)) return bad(ctx, "ry is not 16-byte aligned");
  if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) {
    curve_record rec; int rc = gc_require_ecdsa(ctx, curve, &rec); if (rc != ECSIMD_HIP_OK) return rc;
    if (n == 0) return ECSIMD_HIP_OK;
    return gc_double_scalar_mult(ctx, curve, rec, u1, u2, qx, qy, rx, ry, finite, n, 0);
  }
  REQUIRE_CURVE();
  if (ctx->ref_square) return bad(ctx, "double_scalar_mult is not the reference's algorithm: no ECSIMD_HIP_REF_SQUARE_COMPAT form");
  if (n == 0) return ECSIMD_HIP_OK;
  return double_scalar_mult_impl(ctx, curve, u1, u2, qx, qy, rx, ry, finite, n, 0); }

// ECDSA's acceptance test on top of double_scalar_mult: ok[i] = Q[i] is a valid public key && u1*G + u2*Q is finite && its x mod n == r[i].
// `extra` bytes are kept untouched at the end of the workspace for the caller (ecdsa_verify's u1, u2 and range flags).
namespace {
struct verify_layout { size_t front, behind; };
verify_layout verify_sizes(size_t n) {
  const size_t chunk = n < VARWIN_CHUNK ? n : VARWIN_CHUNK;
  return {7 * chunk * 32 + launch::varwin_scratch_bytes(chunk), n * 32 + ((n + 15) / 16) * 16};
}
int ecdsa_verify_rx_impl(ecsimd_hip_ctx* ctx, int curve, const uint64_t* u1, const uint64_t* u2, const uint64_t* qx, const uint64_t* qy,
                         const uint64_t* r, uint8_t* ok, size_t n, size_t extra) {
  (void)hipSetDevice(ctx->device);
  // x coordinates and the finite flags of the sums live behind double_scalar_mult's own workspace use
  const verify_layout L = verify_sizes(n);
  // sizes the workspace (and builds the table) first, so that the pointers taken below stay valid
  int rc = ensure_window_table(ctx, curve, (ctx->window16_table[curve] != nullptr || n >= BIG_TABLE_WORTH_IT) ? launch::BIG_WINDOW_BITS : SIGNED_WBITS);
  if (rc == ECSIMD_HIP_OK) rc = ensure_workspace(ctx, L.front + L.behind + extra);
  if (rc != ECSIMD_HIP_OK) return rc;
  uint64_t* rx = ctx->workspace + L.front / 8;
  uint8_t* fin = reinterpret_cast<uint8_t*>(rx + 4 * n);
  rc = double_scalar_mult_impl(ctx, curve, u1, u2, qx, qy, rx, nullptr, fin, n, L.behind + extra);
  if (rc != ECSIMD_HIP_OK) return rc;
  launch::x_mod_n_equals(ctx->stream, curve, rx, fin, r, ok, n);       // fin is 0 for the lanes whose Q failed validation
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "ecdsa_verify_rx launch");
}
}  // namespace

int ecsimd_hip_ecdsa_verify_rx(ecsimd_hip_ctx* ctx, int curve, const uint64_t* u1, const uint64_t* u2, const uint64_t* qx, const uint64_t* qy,
                               const uint64_t* r, uint8_t* ok, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(u1); REQUIRE_PTR(u2); REQUIRE_PTR(qx); REQUIRE_PTR(qy); REQUIRE_PTR(r);
  if (!ok && n) return bad(ctx, "ok is null");
  if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) {
    curve_record rec; int rc = gc_require_ecdsa(ctx, curve, &rec); if (rc != ECSIMD_HIP_OK) return rc;
    if (n == 0) return ECSIMD_HIP_OK;
    (void)hipSetDevice(ctx->device);
    return gc_ecdsa_verify_rx(ctx, curve, rec, u1, u2, qx, qy, r, ok, n, 0);
  }
  REQUIRE_CURVE();
  if (ctx->ref_square) return bad(ctx, "ecdsa_verify_rx is not the reference's algorithm: no ECSIMD_HIP_REF_SQUARE_COMPAT form");
  if (n == 0) return ECSIMD_HIP_OK;
  return ecdsa_verify_rx_impl(ctx, curve, u1, u2, qx, qy, r, ok, n, 0); }

// The whole verification (SEC 1 v2 4.1.4, FIPS 186-5 6.4.2): range checks and u1 = e / s, u2 = r / s modulo the group order on the device
// (k_gfield.hip k_ecdsa_scalars: one shared inversion per up to 128 signatures), then the acceptance test above.
int ecsimd_hip_ecdsa_verify(ecsimd_hip_ctx* ctx, int curve, const uint64_t* e, const uint64_t* r, const uint64_t* s_, const uint64_t* qx, const uint64_t* qy,
                            uint8_t* ok, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(e); REQUIRE_PTR(r); REQUIRE_PTR(s_); REQUIRE_PTR(qx); REQUIRE_PTR(qy);
  if (!ok && n) return bad(ctx, "ok is null");
  if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) {
    // the same verification on a registered curve: u1, u2 modulo ITS order (the record's field of n), then two ladders, an affine addition, x mod n = r
    curve_record rec; int rc = gc_require_ecdsa(ctx, curve, &rec); if (rc != ECSIMD_HIP_OK) return rc;
    if (n == 0) return ECSIMD_HIP_OK;
    if (n > (size_t)0x7fffffff * BLOCK) return bad(ctx, "batch too large");
    (void)hipSetDevice(ctx->device);
    const gc_layout L0 = gc_plan(nullptr, n, gc_window_possible(rec));
    const size_t behind = n * 32 + ((n + 15) / 16) * 16, extra = 2 * n * 32 + ((n + 15) / 16) * 16;
    rc = ensure_workspace(ctx, L0.bytes + behind + extra);
    if (rc != ECSIMD_HIP_OK) return rc;
    uint64_t* u1 = ctx->workspace + (L0.bytes + behind) / 8; uint64_t* u2 = u1 + 4 * n;
    uint8_t* in_range = reinterpret_cast<uint8_t*>(u2 + 4 * n);
    launch::ecdsa_scalars(ctx->stream, rec.N, e, r, s_, u1, u2, in_range, n);
    rc = gc_ecdsa_verify_rx(ctx, curve, rec, u1, u2, qx, qy, r, ok, n, extra);
    if (rc != ECSIMD_HIP_OK) return rc;
    launch::mask_op(ctx->stream, ECSIMD_HIP_MASK_AND, ok, in_range, ok, n);
    hipError_t err = hipGetLastError();
    return err == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, err, "ecdsa_verify (registered curve) launch");
  }
  REQUIRE_CURVE();
  if (ctx->ref_square) return bad(ctx, "ecdsa_verify is not the reference's algorithm: no ECSIMD_HIP_REF_SQUARE_COMPAT form");
  if (n == 0) return ECSIMD_HIP_OK;
  if (n > (size_t)0x7fffffff * BLOCK) return bad(ctx, "batch too large");
  (void)hipSetDevice(ctx->device);
  gmod N; if (!lookup_modulus(curve == ECSIMD_HIP_P256 ? ECSIMD_HIP_FIELD_P256_ORDER : ECSIMD_HIP_FIELD_SECP256K1_ORDER, &N)) return bad(ctx, "group order missing from the registry");
  const verify_layout L = verify_sizes(n);
  const size_t extra = 2 * n * 32 + ((n + 15) / 16) * 16;        // u1, u2, range flags
  int rc = ensure_window_table(ctx, curve, (ctx->window16_table[curve] != nullptr || n >= BIG_TABLE_WORTH_IT) ? launch::BIG_WINDOW_BITS : SIGNED_WBITS);
  if (rc == ECSIMD_HIP_OK) rc = ensure_workspace(ctx, L.front + L.behind + extra);
  if (rc != ECSIMD_HIP_OK) return rc;
  uint64_t* u1 = ctx->workspace + (L.front + L.behind) / 8;
  uint64_t* u2 = u1 + 4 * n;
  uint8_t* in_range = reinterpret_cast<uint8_t*>(u2 + 4 * n);
  launch::ecdsa_scalars(ctx->stream, N, e, r, s_, u1, u2, in_range, n);
  rc = ecdsa_verify_rx_impl(ctx, curve, u1, u2, qx, qy, r, ok, n, extra);     // the workspace is already this large: nothing moves
  if (rc != ECSIMD_HIP_OK) return rc;
  launch::mask_op(ctx->stream, ECSIMD_HIP_MASK_AND, ok, in_range, ok, n);      // u1 = u2 = 0 already fails (the sum is infinite); the flag says why
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, err, "ecdsa_verify launch"); }

// Signing: R = k G on the constant-time comb (the kernel behind ALG_WINDOWED | ALG_CONSTANT_TIME), x only; then the scalar-field half.
int ecsimd_hip_ecdsa_sign(ecsimd_hip_ctx* ctx, int curve, const uint64_t* e, const uint64_t* d, const uint64_t* k, uint64_t* r, uint64_t* s_, uint8_t* ok, size_t n) {
  REQUIRE_CTX(); REQUIRE_PTR(e); REQUIRE_PTR(d); REQUIRE_PTR(k); REQUIRE_PTR(r); REQUIRE_PTR(s_);
  if (!ok && n) return bad(ctx, "ok is null");
  if (curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE) {
    // signing on a registered curve: k G from the constant-time comb of its generator (k_gcomb.hip; n < 2^255: through the reference's ladder, constant-time as it
    // is -- tests/test_constant_time_isa.py holds both to the ISA), x by the select-only shared inversion, then the scalar-field kernel with the record's order
    curve_record rec; int rc = gc_require_ecdsa(ctx, curve, &rec); if (rc != ECSIMD_HIP_OK) return rc;
    if (overlaps(r, e) || overlaps(r, d) || overlaps(r, k) || overlaps(s_, e) || overlaps(s_, d) || overlaps(s_, k) || overlaps(r, s_)) return bad(ctx, "r and s must not alias an input or each other");
    if (n == 0) return ECSIMD_HIP_OK;
    if (n > (size_t)0x7fffffff * BLOCK) return bad(ctx, "batch too large");
    (void)hipSetDevice(ctx->device);
    gc_layout L = gc_plan(nullptr, n);
    if (L.chunk != n) return bad(ctx, "ecdsa_sign on a registered curve: at most 2^22 signatures per call");
    rc = ensure_workspace(ctx, L.bytes);
    if (rc != ECSIMD_HIP_OK) return rc;
    const uint32_t* comb = nullptr;                             // k G from the generator's constant-time comb (5-bit windows, every entry of a window read), where the curve has one
    if (gc_comb_possible(rec)) { rc = ensure_gc_comb(ctx, curve, rec, &comb, 5); if (rc != ECSIMD_HIP_OK) { if (!capturing(ctx)) return rc; comb = nullptr; } }   // (no table yet and a capture in progress: the ladder)
    L = gc_plan(ctx->workspace, n);
    if (comb) {
      launch::gc_base_windowed_s(ctx->stream, rec.G, order_words(rec), 5, k, comb, L.j[0], L.j[1], L.j[2], n);
      launch::gc_to_affine_batched(ctx->stream, rec.G, L.j[0], L.j[1], L.j[2], L.gx, nullptr, n);
    } else gc_safe_mult(ctx->stream, rec, L, L.adj1, L.neg1, k, nullptr, nullptr, L.gx, nullptr, n);  // x(k G): the negation of a degenerate nonce's product does not touch x
    launch::ecdsa_sign_scalars(ctx->stream, rec.N, e, d, k, L.gx, r, s_, ok, n);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipMemsetAsync(ctx->workspace, 0, L.bytes, ctx->stream);              // the nonce's adjusted copy, the Jacobian k G and x: gone before the call returns
    return err == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, err, "ecdsa_sign (registered curve) launch");
  }
  REQUIRE_CURVE();
  if (ctx->ref_square) return bad(ctx, "ecdsa_sign is not the reference's algorithm: no ECSIMD_HIP_REF_SQUARE_COMPAT form");
  if (!(ECS_FIXED4_ODD && ECS_SIGNED_ODD)) return bad(ctx, "this build (-DECS_FIXED4_ODD=0 / -DECS_SIGNED_ODD=0) has no constant-time comb");
  if (overlaps(r, e) || overlaps(r, d) || overlaps(r, k) || overlaps(s_, e) || overlaps(s_, d) || overlaps(s_, k) || overlaps(r, s_)) return bad(ctx, "r and s must not alias an input or each other");
  if (n == 0) return ECSIMD_HIP_OK;
  if (n > (size_t)0x7fffffff * BLOCK) return bad(ctx, "batch too large");
  (void)hipSetDevice(ctx->device);
  gmod N; if (!lookup_modulus(curve == ECSIMD_HIP_P256 ? ECSIMD_HIP_FIELD_P256_ORDER : ECSIMD_HIP_FIELD_SECP256K1_ORDER, &N)) return bad(ctx, "group order missing from the registry");
  const int ctbits = curve == ECSIMD_HIP_P256 ? CT_WBITS : CT_WBITS_SECP;
  if (!ctbits) return bad(ctx, "this build has no 5- / 6-bit constant-time comb");
  int rc = ensure_window_table(ctx, curve, ctbits);
  if (rc == ECSIMD_HIP_OK) rc = ensure_workspace(ctx, 4 * n * 32);
  if (rc != ECSIMD_HIP_OK) return rc;
  uint64_t* jx = ctx->workspace; uint64_t* jy = jx + 4 * n; uint64_t* jz = jy + 4 * n; uint64_t* rx = jz + 4 * n;
  hipStream_t st = ctx->stream;
  launch::base_windowed_signed(st, curve, ctbits, k, ctx->windowct_table[curve], jx, jy, jz, n, true);      // k >= n is reduced by the comb; the lane is refused below
  launch::to_affine_batched(st, curve, jx, jy, jz, rx, nullptr, n, true);
  launch::ecdsa_sign_scalars(st, N, e, d, k, rx, r, s_, ok, n);
  hipError_t err = hipGetLastError();
  // The workspace held the Jacobian k G (X, Y, Z) and its affine x: with the public r, the Z of an unnormalised k G gives bits of the nonce away
  // (the projective-coordinate leak), and the block outlives the call (grow-only, hipFree does not wipe).  Zero it on the same stream, behind the kernels.
  if (err == hipSuccess) err = hipMemsetAsync(ctx->workspace, 0, 4 * n * 32, st);
  return err == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, err, "ecdsa_sign launch"); }

int ecsimd_hip_fe29_raw(ecsimd_hip_ctx* ctx, int curve, int op, const int32_t* in, int32_t* out, size_t n, int swap) {
  REQUIRE_CTX(); if ((!in || !out) && n) return bad(ctx, "fe29_raw: null pointer");
  if (op < 0 || op > launch::RAW_ZADDU) return bad(ctx, "fe29_raw: unknown function");
  gcurve GC; const bool registered = curve >= ECSIMD_HIP_FIRST_REGISTERED_CURVE;
  if (registered) { if (!lookup_curve(curve, &GC)) return bad(ctx, "unknown curve id"); }
  else REQUIRE_CURVE();
  if (n == 0) return ECSIMD_HIP_OK;
  if (n > (size_t)0x7fffffff * BLOCK) return bad(ctx, "batch too large");
  hipError_t e = hipSetDevice(ctx->device); if (e != hipSuccess) return fail(ctx, e, "hipSetDevice");
  if (!launch::fe29_raw(ctx->stream, registered ? 2 : curve, registered ? &GC : nullptr, op, in, out, n, swap ? 0xffffffffu : 0u)) return bad(ctx, "fe29_raw: this function does not exist for this curve");
  e = hipGetLastError();
  return e == hipSuccess ? ECSIMD_HIP_OK : fail(ctx, e, "fe29_raw launch"); }

int ecsimd_hip_workspace_info(ecsimd_hip_ctx* ctx, const void** dptr, size_t* bytes) {
  REQUIRE_CTX(); if (!dptr || !bytes) return bad(ctx, "workspace_info: null output");
  *dptr = ctx->workspace; *bytes = ctx->workspace_bytes; return ECSIMD_HIP_OK; }

int ecsimd_hip_scalar_mult_p256(ecsimd_hip_ctx* ctx, const uint64_t* k, const uint64_t* xm, const uint64_t* ym, uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n) {
  return ecsimd_hip_scalar_mult(ctx, ECSIMD_HIP_P256, k, xm, ym, ox, oy, oz, n, ECSIMD_HIP_BASE_MGRY | ECSIMD_HIP_OUT_JACOBIAN); }

int ecsimd_hip_fill_random(ecsimd_hip_ctx* ctx, uint64_t* out, size_t n, uint64_t seed, uint64_t stream, uint64_t first_index, int clear_top_bits) {
  REQUIRE_CTX(); REQUIRE_PTR(out); if (clear_top_bits < 0 || clear_top_bits > 63) return bad(ctx, "clear_top_bits");
  RUN(launch::fill_random(s, out, n, seed, stream, first_index, clear_top_bits)); }

int ecsimd_hip_peak_mad32(ecsimd_hip_ctx* ctx, int iters, double* mads, double* ms) {
  REQUIRE_CTX(); if (iters < 1 || !mads || !ms) return bad(ctx, "peak_mad32 arguments");
  (void)hipSetDevice(ctx->device);
  const int blocks = ctx->cus * 8;          // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch::peak_mad32(ctx->stream, blocks, ctx->sink, 16, 1u);   // warm-up
  (void)hipEventRecord(e0, ctx->stream);
  launch::peak_mad32(ctx->stream, blocks, ctx->sink, iters, 2u);
  (void)hipEventRecord(e1, ctx->stream);
  hipError_t e = hipEventSynchronize(e1);
  float t = 0; if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (e != hipSuccess) return fail(ctx, e, "peak_mad32");
  *ms = t; *mads = (double)blocks * BLOCK * (double)iters * launch::PEAK_MADS_PER_LANE_PER_ITER;
  return ECSIMD_HIP_OK; }

}  // extern "C"
