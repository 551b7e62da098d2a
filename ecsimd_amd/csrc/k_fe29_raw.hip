// k_fe29_raw.hip -- ONE function of the reduced-radix layer (fe29.cuh) on RAW operands: nine signed 32-bit limbs per coordinate, exactly as the loops hold
// them between two operations (ecsimd_hip_fe29_raw; round 5).  A diagnostic entry point: it lets a test hand the device the states
// and operand pairs at which the interval proofs of tools/radix29_model.py reach their largest columns (tests/golden/fe29_witnesses.json) -- inputs the
// other entry points cannot produce, since their operands enter the loops as tight limbs -- and compare the result limb for limb with the exact model.
// Layout: element e's coordinate c, limb l at in[(e * NIN + c) * 9 + l]; outputs likewise with NOUT.
#include "kernels.h"
#include "gcurve.cuh"

namespace ecsimd_hip {
namespace {
using launch::BLOCK;
ECS_DEV fe29 ld29(const int32_t* p) { fe29 r;
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) r.l[i] = p[i];
  return r; }
ECS_DEV void st29(int32_t* p, const fe29& v) {
#pragma unroll
  for (int i = 0; i < R29_LIMBS; ++i) p[i] = v.l[i];
}

template <int C, int OP> __global__ void __launch_bounds__(BLOCK) k_fe29_raw(r29_ctx<C> cx, const int32_t* __restrict__ in, int32_t* __restrict__ out, size_t n, uint32_t swap) {
  const size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (e >= n) return;
  constexpr int NIN = launch::fe29_raw_inputs(OP), NOUT = launch::fe29_raw_outputs(OP);
  const int32_t* a = in + e * NIN * R29_LIMBS;
  int32_t* o = out + e * NOUT * R29_LIMBS;
  auto I = [&](int c) { return ld29(a + c * R29_LIMBS); };
  auto O = [&](int c, const fe29& v) { st29(o + c * R29_LIMBS, v); };
  if constexpr (OP == launch::RAW_ZDAU) {
    coz29 s{I(0), I(1), I(2), I(3), I(4), I(5)};                              // x1, x2, dx, y1, dy, z
    zdau29<C>(s, swap, cx);
    O(0, s.x1); O(1, s.x2); O(2, s.dx); O(3, s.y1); O(4, s.dy); O(5, s.z);
  } else if constexpr (OP == launch::RAW_MUL) {
    O(0, mul29<C>(I(0), I(1), cx));
  } else if constexpr (OP == launch::RAW_SQR) {
    O(0, sqr29<C>(I(0), cx));
  } else if constexpr (OP == launch::RAW_ZADDU) {                             // x1, y1, x2, y2, z -> rx, ry, x1', y1', z', dx
    fe29 x1 = I(0), y1 = I(1), z = I(4), rx, ry, dx;
    zaddu29<C>(x1, y1, I(2), I(3), z, rx, ry, dx, cx);
    O(0, rx); O(1, ry); O(2, x1); O(3, y1); O(4, z); O(5, dx);
  } else if constexpr (r29_prime<C>::dense) {                                 // a registered curve: the generator's comb (madd29) and the window loop (gjdbl29, dbl_add29) with the dense reduction
    const jpoint29 P{I(0), I(1), I(2)};
    jpoint29 R = P;
    if constexpr (OP == launch::RAW_MADD) R = madd29<C>(P, I(3), I(4), cx);
    else if constexpr (OP == launch::RAW_DBL_ADD) R = dbl_add29<C>(P, I(3), I(4), cx);
    else if constexpr (OP == launch::RAW_GJDBL) {                             // X, Y, Z, W -> X, Y, Z, W; swap: the doubled point's W is formed (WOUT)
      fe29 w = I(3);
      if (swap) R = gjdbl29<C, true>(P, w, cx); else R = gjdbl29<C, false>(P, w, cx);
      O(3, w);
    }
    O(0, R.x); O(1, R.y); O(2, R.z);
  } else {                                                                    // the window kernels' functions exist for the two built-in primes
    const jpoint29 P{I(0), I(1), I(2)};
    jpoint29 R;
    if constexpr (OP == launch::RAW_MADD) R = madd29<C>(P, I(3), I(4));
    else if constexpr (OP == launch::RAW_JDBL) R = jdbl29<C>(P);
    else if constexpr (OP == launch::RAW_DBL_ADD) R = dbl_add29<C>(P, I(3), I(4));
    else if constexpr (OP == launch::RAW_MADDV) { fe29 H, r; madd29_hr<C>(P, I(3), I(4), H, r); R = madd29v_finish<C>(P, H, r); }
    else if constexpr (!r29_prime<C>::p256 && OP == launch::RAW_PDBL) R = pdbl29<C>(P);
    else if constexpr (!r29_prime<C>::p256 && OP == launch::RAW_PADD) R = padd29<C>(P, I(3), I(4));
    else R = P;
    O(0, R.x); O(1, R.y); O(2, R.z);
  }
}
}  // namespace

namespace launch {
template <int C> static bool raw_dispatch(hipStream_t s, const r29_ctx<C>& cx, int op, const int32_t* in, int32_t* out, size_t n, uint32_t swap) {
#define CASE(OP) case OP: hipLaunchKernelGGL((k_fe29_raw<C, OP>), grid_for(n), dim3(BLOCK), 0, s, cx, in, out, n, swap); return true
  switch (op) {
    CASE(RAW_ZDAU); CASE(RAW_MUL); CASE(RAW_SQR);
    default: break;
  }
  if constexpr (r29_prime<C>::dense) {
    switch (op) { CASE(RAW_MADD); CASE(RAW_DBL_ADD); CASE(RAW_GJDBL); CASE(RAW_ZADDU); default: break; }
  } else {
    switch (op) {
      CASE(RAW_MADD); CASE(RAW_JDBL); CASE(RAW_DBL_ADD); CASE(RAW_MADDV);
      default: break;
    }
    if constexpr (!r29_prime<C>::p256) {
      switch (op) { CASE(RAW_PDBL); CASE(RAW_PADD); CASE(RAW_ZADDU); default: break; }
    }
  }
#undef CASE
  return false;
}
// curve: 0 P-256, 1 secp256k1 (the loops' own domain: fe29.cuh CURVE_SECP256K1_CLASSICAL), 2 a registered curve (G != nullptr).  false: no such function there.
bool fe29_raw(hipStream_t s, int curve, const gcurve* G, int op, const int32_t* in, int32_t* out, size_t n, uint32_t swap) {
  if (curve == 0) return raw_dispatch<CURVE_P256>(s, r29_ctx<CURVE_P256>{}, op, in, out, n, swap);
  if (curve == 1) return raw_dispatch<CURVE_SECP256K1_CLASSICAL>(s, r29_ctx<CURVE_SECP256K1_CLASSICAL>{}, op, in, out, n, swap);
  return G != nullptr && raw_dispatch<CURVE_GENERIC>(s, G->r29, op, in, out, n, swap);
}
}  // namespace launch
}  // namespace ecsimd_hip
