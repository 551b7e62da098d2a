// scalar_mult_p256_adapter.h -- what a caller of the reference-side binding includes.
//
// The reference exports ONE function, scalar_mult_p256(x, P) over one eve::wide of four lanes (lib/scalar_mult_p256.cpp:10-12); its return
// type is deduced, so no header of the reference declares it.  This header declares that function with its real type and, beside it, the
// batch form a GPU needs: spans of wides, staged once, ONE launch.  Compiled with the reference's own headers on the include path.
#pragma once
#include <ecsimd/curve_group.h>
#include <ecsimd/curve_nist_p256.h>
#include <ecsimd_hip.h>

#include <span>

namespace ecsimd_mi355x {
using Curve = ecsimd::curve_nist_p256;
using WBN   = ecsimd::curve_wide_bn_t<Curve>;
using WJCP  = ecsimd::wide_jacobian_curve_point<Curve>;
}  // namespace ecsimd_mi355x

// lib/scalar_mult_p256.cpp:10-12, same meaning: lane l of the result = x[l] * P[l]; P.z must be mgry(1); Jacobian Montgomery out
ecsimd_mi355x::WJCP scalar_mult_p256(ecsimd_mi355x::WBN const& x, ecsimd_mi355x::WJCP const& P);
// The batch form: out[w] = scalar_mult_p256(x[w], P[w]) for every wide, 4 * x.size() lanes in ONE launch (the three spans have one length).
void scalar_mult_p256(std::span<const ecsimd_mi355x::WBN> x, std::span<const ecsimd_mi355x::WJCP> P, std::span<ecsimd_mi355x::WJCP> out);
// true when the spans travel as raw bytes and the lane transposition runs on the device (the layout check passed)
bool scalar_mult_p256_transposes_on_the_device();
// ECSIMD_HIP_REF_SQUARE_COMPAT for every launch of the adapter (= ecsimd_hip_set_ref_square_compat on the context below; large batches run on a second context
// as well, which takes the option over from this one at every call)
void scalar_mult_p256_set_ref_square_compat(bool on);
// The context the adapter runs on (created on first use, device 0): its options are the adapter's.
ecsimd_hip_ctx* scalar_mult_p256_context();
