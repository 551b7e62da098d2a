#include "field.cuh"
#define ECS_CURVE ecsimd_hip::CURVE_P256
#define ECS_LADDER_X 1          // the x-coordinate-only ladder (a != 0 curves)
#include "k_ladder.inc"
