"""Flag values of include/ecsimd_hip.h (scalar_mult `flags`)."""
BASE_CLASSICAL = 0
BASE_MGRY = 1
OUT_JACOBIAN = 0
OUT_AFFINE = 2
ALG_WINDOWED = 4
ALG_WINDOWED_SIGNED = 8
ALG_NO_ENDOMORPHISM = 16
