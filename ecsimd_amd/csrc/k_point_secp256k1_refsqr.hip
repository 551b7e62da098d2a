#include "field.cuh"
#define ECS_CURVE ecsimd_hip::CURVE_SECP256K1_REFSQR
#include "k_point.inc"
