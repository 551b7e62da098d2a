"""Flag values of include/ecsimd_hip.h (scalar_mult `flags`)."""
BASE_CLASSICAL = 0
BASE_MGRY = 1
OUT_JACOBIAN = 0
OUT_AFFINE = 2
ALG_WINDOWED = 4
ALG_WINDOWED_SIGNED = 8
ALG_NO_ENDOMORPHISM = 16
ALG_WINDOWED_BIG = 32
REF_SQUARE_COMPAT = 64
ALG_CONSTANT_TIME = 128       # scalar_mult_base + ALG_WINDOWED: every table entry read, the wanted one kept under lane masks
BASE_GENERATOR = 512          # ecsimd_hip_scalar_mult with x = y = NULL: the base point is the generator
LADDER_RADIX32 = 256          # ladder only: the loop on 8 x 32-bit canonical words (rounds 1-3) instead of nine signed 29-bit limbs
GROUP_NO_GATHER = 0x10000    # ecsimd_hip_group_scalar_mult only: compute without the exchange
