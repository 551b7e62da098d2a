// k_field.hip -- GF(p) element-wise kernels for both curves (reference layer L3: modular.h,
// mgry_mul.h, mgry_ops.h, mgry.h, gfp.h).
#include "kernels.h"
#include "point.cuh"

namespace ecsimd_hip {
namespace {
using launch::BLOCK;
using launch::field_op;
#define GID size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= n) return

template <int C, int OP> __global__ void __launch_bounds__(BLOCK) k_binop(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  GID; const fe x = fe_load(a, i), y = fe_load(b, i); fe r;
  if constexpr (OP == launch::F_MOD_ADD) r = fe_add<C>(x, y);
  else if constexpr (OP == launch::F_MOD_SUB) r = fe_sub<C>(x, y);
  else r = fe_mul<C>(x, y);
  fe_store(out, i, r);
}
template <int C, int OP> __global__ void __launch_bounds__(BLOCK) k_unop(const uint64_t* a, uint64_t* out, size_t n) {
  GID; const fe x = fe_load(a, i); fe r;
  if constexpr (OP == launch::F_MGRY_SQR) r = fe_sqr<C>(x);
  else if constexpr (OP == launch::F_FROM_CLASSICAL) r = fe_from_classical<C>(x);
  else if constexpr (OP == launch::F_TO_CLASSICAL) r = fe_to_classical<C>(x);
  else if constexpr (OP == launch::F_INVERSE) r = from_fast<C>(fe_inverse<curve_domain<C>::fast>(to_fast<C>(x)));
  else r = fe_opposite<C>(x);
  fe_store(out, i, r);
}
// classical a*b mod p (an extension: the reference has mod_add / mod_sub but no mod_mul).
// secp256k1: one multiply + pseudo-Mersenne reduction; P-256: two Montgomery multiplies (ab/R, then *R^2/R).
template <int C> __global__ void __launch_bounds__(BLOCK) k_mod_mul(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  GID; const fe x = fe_load(a, i), y = fe_load(b, i);
  constexpr int CI = curve_domain<C>::fast;
  if constexpr (CI == C) fe_store(out, i, fe_mul<C>(fe_mul<C>(x, y), FE_CONST(C, RSQ)));
  else fe_store(out, i, fe_mul<CI>(x, y));
}
template <int C> __global__ void __launch_bounds__(BLOCK) k_shift_left(const uint64_t* a, int count, uint64_t* out, size_t n) {
  GID; fe x = fe_load(a, i);
  int left = count & 0xff;
  // ECSIMD_HIP_SHIFT_FUSED: pairs of doublings as one quadrupling (field.cuh fe_shl2, what the point formulas use on their own
  // canonical intermediates): the same residue for a < p; for a >= p only the literal doubling chain is the reference's value
  if (count & 0x100) for (; left >= 2; left -= 2) x = fe_shl2<C>(x);
  for (; left > 0; --left) x = fe_dbl<C>(x);
  fe_store(out, i, x);
}
template <int C> __global__ void __launch_bounds__(BLOCK) k_reduce(const uint64_t* a8, uint64_t* out, size_t n) {
  GID; fe2 t = fe2_load(a8, i); fe_store(out, i, mgry_reduce<C>(t));
}
template <int C> __global__ void __launch_bounds__(BLOCK) k_pow(const uint64_t* a, launch::words8 e, uint64_t* out, size_t n) {
  GID; fe_store(out, i, from_fast<C>(fe_pow<curve_domain<C>::fast>(to_fast<C>(fe_load(a, i)), e.w)));
}
template <int C> __global__ void __launch_bounds__(BLOCK) k_sqrt(const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n) {
  GID; const fe x = fe_load(a, i);
  constexpr int CI = curve_domain<C>::fast;
  const fe xf = to_fast<C>(x);
  const fe s = fe_sqrt_candidate<CI>(xf);              // gfp.h:46-54
  fe_store(out, i, from_fast<C>(s)); if (ok) ok[i] = (uint8_t)fe_eq(fe_sqr<CI>(s), xf);
}
}  // namespace

namespace launch {
#define GO(kern, ...) hipLaunchKernelGGL(kern, grid_for(n), dim3(BLOCK), 0, s, __VA_ARGS__)
// curve: the two API curves and their ECSIMD_HIP_REF_SQUARE_COMPAT instances (field.cuh)
#define BY_CURVE(kern, ...) do { switch (curve) { case CURVE_P256: GO((kern<CURVE_P256>), __VA_ARGS__); break; case CURVE_SECP256K1: GO((kern<CURVE_SECP256K1>), __VA_ARGS__); break; \
    case CURVE_P256_REFSQR: GO((kern<CURVE_P256_REFSQR>), __VA_ARGS__); break; default: GO((kern<CURVE_SECP256K1_REFSQR>), __VA_ARGS__); break; } } while (0)
#define BY_CURVE_OP(kern, OP, ...) do { switch (curve) { case CURVE_P256: GO((kern<CURVE_P256, OP>), __VA_ARGS__); break; case CURVE_SECP256K1: GO((kern<CURVE_SECP256K1, OP>), __VA_ARGS__); break; \
    case CURVE_P256_REFSQR: GO((kern<CURVE_P256_REFSQR, OP>), __VA_ARGS__); break; default: GO((kern<CURVE_SECP256K1_REFSQR, OP>), __VA_ARGS__); break; } } while (0)

void field_binop(hipStream_t s, int curve, field_op op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  switch (op) {
    case F_MOD_ADD: BY_CURVE_OP(k_binop, F_MOD_ADD, a, b, out, n); break;
    case F_MOD_SUB: BY_CURVE_OP(k_binop, F_MOD_SUB, a, b, out, n); break;
    default: BY_CURVE_OP(k_binop, F_MGRY_MUL, a, b, out, n); break;
  }
}
void field_unop(hipStream_t s, int curve, field_op op, const uint64_t* a, uint64_t* out, size_t n) {
  switch (op) {
    case F_MGRY_SQR: BY_CURVE_OP(k_unop, F_MGRY_SQR, a, out, n); break;
    case F_FROM_CLASSICAL: BY_CURVE_OP(k_unop, F_FROM_CLASSICAL, a, out, n); break;
    case F_TO_CLASSICAL: BY_CURVE_OP(k_unop, F_TO_CLASSICAL, a, out, n); break;
    case F_INVERSE: BY_CURVE_OP(k_unop, F_INVERSE, a, out, n); break;
    default: BY_CURVE_OP(k_unop, F_OPPOSITE, a, out, n); break;
  }
}
void mod_mul(hipStream_t s, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { BY_CURVE(k_mod_mul, a, b, out, n); }
void mod_shift_left(hipStream_t s, int curve, const uint64_t* a, int count, uint64_t* out, size_t n) { BY_CURVE(k_shift_left, a, count, out, n); }
void mgry_reduce(hipStream_t s, int curve, const uint64_t* a8, uint64_t* out, size_t n) { BY_CURVE(k_reduce, a8, out, n); }
void mgry_pow(hipStream_t s, int curve, const uint64_t* a, const words8& e, uint64_t* out, size_t n) { BY_CURVE(k_pow, a, e, out, n); }
void gfp_sqrt(hipStream_t s, int curve, const uint64_t* a, uint64_t* out, uint8_t* ok, size_t n) { BY_CURVE(k_sqrt, a, out, ok, n); }
}  // namespace launch
}  // namespace ecsimd_hip
