// ecsimd/scalar_mult_p256.h -- the reference's one exported entry point (lib/scalar_mult_p256.cpp:10-12),
// declared in a header here so that callers can actually use it.
#ifndef ECSIMD_SCALAR_MULT_P256_H
#define ECSIMD_SCALAR_MULT_P256_H
#include <ecsimd/curve_group.h>
#include <ecsimd/curve_nist_p256.h>

inline ecsimd::wide_jacobian_curve_point<ecsimd::curve_nist_p256>
scalar_mult_p256(ecsimd::curve_wide_bn_t<ecsimd::curve_nist_p256> const& x,
                 ecsimd::wide_jacobian_curve_point<ecsimd::curve_nist_p256> const& P) {
  return ecsimd::curve_group<ecsimd::curve_nist_p256>::scalar_mult(x, P);
}
#endif
