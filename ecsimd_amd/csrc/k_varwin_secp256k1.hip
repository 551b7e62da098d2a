#include "field.cuh"
#define ECS_CURVE ecsimd_hip::CURVE_SECP256K1
#include "k_varwin.inc"
