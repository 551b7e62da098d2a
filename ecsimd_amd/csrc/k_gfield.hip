// k_gfield.hip -- element-wise kernels of the reference's field layer for a RUN-TIME modulus (gfield.cuh), and the scalar-field half of an
// ECDSA verification (u1 = e / s, u2 = r / s modulo the group order n: SEC 1 v2 4.1.4 steps 1, 4, 5; FIPS 186-5 6.4.2).
//
// The modulus travels as a kernel argument (gmod, 236 bytes, wave-uniform -> SGPRs).  REF = the reference's square() as written
// (ECSIMD_HIP_REF_SQUARE_COMPAT): mgry_sqr, mgry_pow and the power-ladder inverse square with mul.h:160-212, dropped carry included.
#include "kernels.h"
#include "gfield.cuh"

namespace ecsimd_hip {
namespace {
using launch::BLOCK;
#define GID size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= n) return

template <int OP> __global__ void __launch_bounds__(BLOCK) k_g_binop(gmod M, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  GID; const fe x = fe_load(a, i), y = fe_load(b, i); fe r;
  if constexpr (OP == launch::F_MOD_ADD) r = g_add(x, y, M);
  else if constexpr (OP == launch::F_MOD_SUB) r = g_sub(x, y, M);
  else r = g_mul(x, y, M);
  fe_store(out, i, r);
}
template <int OP, bool REF> __global__ void __launch_bounds__(BLOCK) k_g_unop(gmod M, const uint64_t* a, uint64_t* out, size_t n) {
  GID; const fe x = fe_load(a, i); fe r;
  if constexpr (OP == launch::F_MGRY_SQR) r = g_sqr<REF>(x, M);
  else if constexpr (OP == launch::F_FROM_CLASSICAL) r = g_from_classical(x, M);
  else if constexpr (OP == launch::F_TO_CLASSICAL) r = g_to_classical(x, M);
  else if constexpr (OP == launch::F_INVERSE) r = g_pow<REF>(x, M.pm2, M);          // gfp.h:42-44 as written: x^(p-2), whatever p is
  else r = g_opposite(x, M);
  fe_store(out, i, r);
}
// prime moduli, exact squaring: the division-step inverse, per element (the in-place form of gfp_inverse)
__global__ void __launch_bounds__(BLOCK) k_g_inverse_divsteps(gmod M, const uint64_t* a, uint64_t* out, size_t n) {
  GID; fe_store(out, i, g_inverse_mgry(fe_load(a, i), M));
}
// classical a * b mod p (the extension ecsimd_hip_mod_mul): ab/R, then * R^2 / R
__global__ void __launch_bounds__(BLOCK) k_g_mod_mul(gmod M, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  GID; fe_store(out, i, g_mul(g_mul(fe_load(a, i), fe_load(b, i), M), g_words(M.rsq), M));
}
__global__ void __launch_bounds__(BLOCK) k_g_shift_left(gmod M, const uint64_t* a, int count, uint64_t* out, size_t n) {
  GID; fe x = fe_load(a, i);
#pragma unroll 1
  for (int left = count & 0xff; left > 0; --left) x = g_dbl(x, M);              // mgry_ops.h:14-22: `count` doublings
  fe_store(out, i, x);
}
__global__ void __launch_bounds__(BLOCK) k_g_reduce(gmod M, const uint64_t* a8, uint64_t* out, size_t n) {
  GID; fe2 t = fe2_load(a8, i); fe_store(out, i, g_reduce(t, M));
}
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_g_pow(gmod M, const uint64_t* a, launch::words8 e, uint64_t* out, size_t n) {
  GID; fe_store(out, i, g_pow<REF>(fe_load(a, i), e.w, M));
}
template <bool REF> __global__ void __launch_bounds__(BLOCK) k_g_sqrt(gmod M, const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n) {
  GID; const fe x = fe_load(a, i);
  const fe s = g_pow<REF>(x, M.psqrt, M);                                       // gfp.h:46-54
  fe_store(out, i, s); if (ok) ok[i] = (uint8_t)fe_eq(g_sqr<REF>(s, M), x);
}

// GFp::inverse over a batch for a PRIME modulus (gfp.h:42-44), Montgomery's simultaneous inversion as k_affine.inc's k_inverse_batched:
// a lane owns m elements `lanes` apart, out[] holds the prefix products on the way up (so out must not alias a).  0 -> 0.
// The operands are whatever the caller hands over -- the reference never rejects a >= p (tests/ops.cpp:232) -- so "zero" is decided on the
// canonical PRODUCT, not on the operand's words: for a prime p and acc != 0, acc * v / R = 0 (mod p) exactly when v = 0 (mod p), whatever multiple
// of a small p the 256-bit v is.  Such an element is left out of the running product (else it would zero the inverses of up to 127 neighbours)
// and gets 0, the value 0^(p-2) has.  Selects only: the operand may be a secret.
__global__ void __launch_bounds__(256) k_g_inverse_batched(gmod M, const uint64_t* __restrict__ a, uint64_t* __restrict__ out, size_t n, size_t lanes, int m) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= lanes) return;
  const fe one = g_words(M.r);
  fe acc = one;
  for (int j = 0; j < m; ++j) {
    const size_t e = (size_t)j * lanes + g;
    if (e >= n) break;
    const fe t = g_mul(acc, fe_load(a, e), M);
    acc = fe_select(g_zero_mask(t), acc, t);
    fe_store(out, e, acc);
  }
  fe inv = g_inverse_mgry(acc, M);
  int last = m - 1;
  while (last >= 0 && (size_t)last * lanes + g >= n) --last;
  for (int j = last; j >= 0; --j) {
    const size_t e = (size_t)j * lanes + g;
    const fe v = fe_load(a, e);
    const fe prev = (j > 0) ? fe_load(out, (size_t)(j - 1) * lanes + g) : one;
    const uint32_t zero = g_zero_mask(g_mul(prev, v, M));        // the same test as on the way up
    fe iv = g_mul(inv, prev, M);
    inv = fe_select(zero, inv, g_mul(inv, v, M));
#pragma unroll
    for (int k = 0; k < 8; ++k) iv.w[k] &= ~zero;
    fe_store(out, e, iv);
  }
}

// ECDSA verification, the arithmetic modulo the group order n (M = n's gmod; every value classical, 4 x u64 LE limbs):
//   valid = 1 <= r < n and 1 <= s < n          (SEC 1 v2 4.1.4 step 1)
//   w = s^-1 mod n;  u1 = e w mod n;  u2 = r w mod n     (steps 4, 5; e = the digest as an integer, ANY 256-bit value: the
//                                                          Montgomery product takes e < 2^256 as it is, the result is canonical)
// One inversion per lane for its m elements (Montgomery's trick) with NO domain conversion: with acc_j = acc_(j-1) * s_j / R the
// plain inverse I_j of acc_j gives  I_j * acc_(j-1) / R = 1 / s_j  and  I_j * s_j / R = I_(j-1)  -- two Montgomery products per
// element on the way down, one on the way up, then w R = w * R^2 / R and the two products by it: 6 per signature + 1/m inversion.
// u1[] carries the prefix products on the way up.  Invalid lanes get u1 = u2 = 0 (whose sum is the point at infinity: rejected).
__global__ void __launch_bounds__(256) k_ecdsa_scalars(gmod M, const uint64_t* __restrict__ ev, const uint64_t* __restrict__ rv, const uint64_t* __restrict__ sv,
                                                       uint64_t* __restrict__ u1, uint64_t* __restrict__ u2, uint8_t* __restrict__ valid, size_t n, size_t lanes, int m) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= lanes) return;
  const fe N = g_words(M.p);
  fe one;
#pragma unroll
  for (int k = 0; k < 8; ++k) one.w[k] = (k == 0) ? 1u : 0u;
  fe acc = one;
  bool first = true;
  for (int j = 0; j < m; ++j) {
    const size_t e = (size_t)j * lanes + g;
    if (e >= n) break;
    fe s = fe_load(sv, e);
    const fe r = fe_load(rv, e);
    const bool ok = !g_is_zero(s) && g_less(s, N) && !g_is_zero(r) && g_less(r, N);
    valid[e] = (uint8_t)ok;
    if (!ok) s = one;
    acc = first ? s : g_mul(acc, s, M);               // acc_0 = s_0 (plain), acc_j = acc_(j-1) s_j / R
    first = false;
    fe_store(u1, e, acc);
  }
  fe inv = g_inverse_plain(acc, M);
  int last = m - 1;
  while (last >= 0 && (size_t)last * lanes + g >= n) --last;
  const fe rsq = g_words(M.rsq);
  for (int j = last; j >= 0; --j) {
    const size_t e = (size_t)j * lanes + g;
    fe s = fe_load(sv, e);
    const bool ok = valid[e] != 0;
    if (!ok) s = one;
    fe w;
    if (j > 0) { w = g_mul(inv, fe_load(u1, (size_t)(j - 1) * lanes + g), M); inv = g_mul(inv, s, M); }
    else w = inv;
    const fe wr = g_mul(w, rsq, M);                   // w R mod n
    fe a = g_mul(fe_load(ev, e), wr, M), b = g_mul(fe_load(rv, e), wr, M);
    if (!ok) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { a.w[k] = 0; b.w[k] = 0; }
    }
    fe_store(u1, e, a);
    fe_store(u2, e, b);
  }
}

// ECDSA signing, the arithmetic modulo the group order (SEC 1 v2 4.1.3 steps 3-6; M = n's gmod): given the digest e, the private key d, the
// nonce k and x = x(k G) (classical, < p < 2n),
//   r = x mod n,   s = k^-1 (e + r d) mod n,   ok = 1 <= d, k < n  and  r != 0  and  s != 0      (r = s = 0 where not ok: a new nonce is the caller's).
// d and k are SECRETS: everything here is selects and the branch-free generic field functions (g_mul, g_add, the division-step inversion, whose
// control flow and addresses do not depend on their operands); the shared inversion multiplies the nonces of one lane's m elements together, which
// reveals nothing outside the lane.  Seven generic Montgomery products per signature + 1/m inversion; s[] carries the prefix products on the way up.
__global__ void __launch_bounds__(256) k_ecdsa_sign_scalars(gmod M, const uint64_t* __restrict__ ev, const uint64_t* __restrict__ dv, const uint64_t* __restrict__ kv,
                                                            const uint64_t* __restrict__ xv, uint64_t* __restrict__ rv, uint64_t* __restrict__ sv,
                                                            uint8_t* __restrict__ okv, size_t n, size_t lanes, int m) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= lanes) return;
  const fe N = g_words(M.p);
  fe one;
#pragma unroll
  for (int j = 0; j < 8; ++j) one.w[j] = (j == 0) ? 1u : 0u;
  auto in_range = [&](const fe& v) { return (uint32_t)(!g_is_zero(v)) & (uint32_t)g_less(v, N); };
  fe acc = one;
  for (int j = 0; j < m; ++j) {
    const size_t e = (size_t)j * lanes + g;
    if (e >= n) break;                                     // (a bound on the public batch length)
    const fe k = fe_load(kv, e);
    const uint32_t good = 0u - in_range(k);
    const fe kk = fe_select(good, k, one);
    acc = (j == 0) ? kk : g_mul(acc, kk, M);               // acc_0 = k_0, acc_j = acc_(j-1) k_j / R  (k_ecdsa_scalars has the algebra)
    fe_store(sv, e, acc);
  }
  fe inv = g_inverse_plain(acc, M);
  int last = m - 1;
  while (last >= 0 && (size_t)last * lanes + g >= n) --last;
  const fe rsq = g_words(M.rsq);
  for (int j = last; j >= 0; --j) {
    const size_t e = (size_t)j * lanes + g;
    const fe k = fe_load(kv, e), d = fe_load(dv, e);
    const uint32_t good = in_range(k) & in_range(d);
    const fe kk = fe_select(0u - in_range(k), k, one);
    fe w;
    if (j > 0) { w = g_mul(inv, fe_load(sv, (size_t)(j - 1) * lanes + g), M); inv = g_mul(inv, kk, M); }
    else w = inv;                                          // k^-1, plain
    fe r = fe_load(xv, e);
    g_cond_sub(r, 0, M);                                   // x < p < 2n: one conditional subtraction reduces it
    fe em = fe_load(ev, e);
    g_cond_sub(em, 0, M);                                  // e < 2^256 < 2n
    const fe rd = g_mul(g_mul(r, d, M), rsq, M);           // r d mod n
    const fe t = g_add(em, rd, M);
    fe s = g_mul(g_mul(w, rsq, M), t, M);                  // (k^-1 R) t / R
    const uint32_t ok = good & (uint32_t)(!g_is_zero(r)) & (uint32_t)(!g_is_zero(s));
    const uint32_t keep = 0u - ok;
#pragma unroll
    for (int q = 0; q < 8; ++q) { r.w[q] &= keep; s.w[q] &= keep; }
    fe_store(rv, e, r); fe_store(sv, e, s);
    okv[e] = (uint8_t)ok;
  }
}
}  // namespace

namespace launch {
#define GO(kern, ...) hipLaunchKernelGGL(kern, grid_for(n), dim3(BLOCK), 0, s, M, __VA_ARGS__)
void gfield_binop(hipStream_t s, const gmod& M, field_op op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  switch (op) {
    case F_MOD_ADD: GO((k_g_binop<F_MOD_ADD>), a, b, out, n); break;
    case F_MOD_SUB: GO((k_g_binop<F_MOD_SUB>), a, b, out, n); break;
    default: GO((k_g_binop<F_MGRY_MUL>), a, b, out, n); break;
  }
}
#define GO_REF(kern, OP, ...) do { if (ref_square) GO((kern<OP, true>), __VA_ARGS__); else GO((kern<OP, false>), __VA_ARGS__); } while (0)
void gfield_unop(hipStream_t s, const gmod& M, field_op op, const uint64_t* a, uint64_t* out, size_t n, bool ref_square) {
  switch (op) {
    case F_MGRY_SQR: GO_REF(k_g_unop, F_MGRY_SQR, a, out, n); break;
    case F_FROM_CLASSICAL: GO((k_g_unop<F_FROM_CLASSICAL, false>), a, out, n); break;
    case F_TO_CLASSICAL: GO((k_g_unop<F_TO_CLASSICAL, false>), a, out, n); break;
    case F_INVERSE:
      // a prime modulus with exact squaring: the division steps (the same unique inverse); anything else: x^(p-2) as gfp.h:42-44 writes it
      if ((M.flags & GMOD_PRIME) && !ref_square) GO(k_g_inverse_divsteps, a, out, n);
      else GO_REF(k_g_unop, F_INVERSE, a, out, n);
      break;
    default: GO((k_g_unop<F_OPPOSITE, false>), a, out, n); break;
  }
}
void gfield_mod_mul(hipStream_t s, const gmod& M, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { GO(k_g_mod_mul, a, b, out, n); }
void gfield_shift_left(hipStream_t s, const gmod& M, const uint64_t* a, int count, uint64_t* out, size_t n) { GO(k_g_shift_left, a, count, out, n); }
void gfield_reduce(hipStream_t s, const gmod& M, const uint64_t* a8, uint64_t* out, size_t n) { GO(k_g_reduce, a8, out, n); }
void gfield_pow(hipStream_t s, const gmod& M, const uint64_t* a, const words8& e, uint64_t* out, size_t n, bool ref_square) {
  if (ref_square) GO(k_g_pow<true>, a, e, out, n); else GO(k_g_pow<false>, a, e, out, n); }
void gfield_sqrt(hipStream_t s, const gmod& M, const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n, bool ref_square) {
  if (ref_square) GO(k_g_sqrt<true>, a, out, ok, n); else GO(k_g_sqrt<false>, a, out, ok, n); }
#undef GO
constexpr size_t G_BATCH_MAX = 128;          // elements that share one inversion
static void batch_shape(size_t n, size_t& lanes, size_t& m) {
  m = n >> 17; if (m < 1) m = 1; if (m > G_BATCH_MAX) m = G_BATCH_MAX;
  lanes = (n + m - 1) / m;
}
void gfield_inverse_batched(hipStream_t s, const gmod& M, const uint64_t* a, uint64_t* out, size_t n) {
  size_t lanes, m; batch_shape(n, lanes, m);
  hipLaunchKernelGGL(k_g_inverse_batched, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, M, a, out, n, lanes, (int)m);
}
void ecdsa_scalars(hipStream_t s, const gmod& M, const uint64_t* e, const uint64_t* r, const uint64_t* sg, uint64_t* u1, uint64_t* u2, uint8_t* valid, size_t n) {
  size_t lanes, m; batch_shape(n, lanes, m);
  hipLaunchKernelGGL(k_ecdsa_scalars, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, M, e, r, sg, u1, u2, valid, n, lanes, (int)m);
}
void ecdsa_sign_scalars(hipStream_t s, const gmod& M, const uint64_t* e, const uint64_t* d, const uint64_t* k, const uint64_t* x, uint64_t* r, uint64_t* sg, uint8_t* ok, size_t n) {
  size_t lanes, m; batch_shape(n, lanes, m);
  hipLaunchKernelGGL(k_ecdsa_sign_scalars, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, M, e, d, k, x, r, sg, ok, n, lanes, (int)m);
}
}  // namespace launch
}  // namespace ecsimd_hip
