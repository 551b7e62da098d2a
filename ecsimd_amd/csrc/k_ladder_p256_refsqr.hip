#include "field.cuh"
#define ECS_CURVE ecsimd_hip::CURVE_P256_REFSQR
#include "k_ladder.inc"
