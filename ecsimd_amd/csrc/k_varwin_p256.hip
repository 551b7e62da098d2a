#include "field.cuh"
#define ECS_CURVE ecsimd_hip::CURVE_P256
#include "k_varwin.inc"
