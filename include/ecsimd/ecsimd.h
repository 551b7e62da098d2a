// ecsimd/ecsimd.h -- umbrella header.
#ifndef ECSIMD_ECSIMD_H
#define ECSIMD_ECSIMD_H
#include <ecsimd/add.h>
#include <ecsimd/bignum.h>
#include <ecsimd/cmp.h>
#include <ecsimd/curve.h>
#include <ecsimd/curve_group.h>
#include <ecsimd/curve_nist_p256.h>
#include <ecsimd/curve_point.h>
#include <ecsimd/curve_point_ops.h>
#include <ecsimd/curve_secp256k1.h>
#include <ecsimd/gfp.h>
#include <ecsimd/ifelse.h>
#include <ecsimd/jacobian_curve_point.h>
#include <ecsimd/literals.h>
#include <ecsimd/mgry.h>
#include <ecsimd/mgry_ops.h>
#include <ecsimd/modular.h>
#include <ecsimd/mul.h>
#include <ecsimd/scalar_mult_p256.h>
#include <ecsimd/sec1.h>
#include <ecsimd/serialization.h>
#include <ecsimd/shift.h>
#include <ecsimd/sub.h>
#include <ecsimd/swap.h>
#include <ecsimd/utility.h>
#endif
