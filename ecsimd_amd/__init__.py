"""ecsimd_amd -- MI355X (gfx950) batched elliptic-curve engine: Python plumbing over the C ABI.

The product is ``libecsimd_hip.so`` (hand-written HIP kernels + the C ABI of ``include/ecsimd_hip.h``)
and the C++ headers in ``include/ecsimd/``.  This package only loads the library with ctypes and
passes device pointers of torch tensors (torch is plumbing: device memory, streams,
``torch.distributed``).  There is NO CPU fallback: importing works anywhere, but creating an
``Engine`` fails loudly when the library or a gfx950 device is missing.
"""
from .engine import Engine, DeviceGroup, shard_range_c, EcsimdHipError, CURVES, P256, SECP256K1, lib_path, load_library  # noqa: F401
from .flags import BASE_CLASSICAL, BASE_MGRY, OUT_JACOBIAN, OUT_AFFINE, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_NO_ENDOMORPHISM, ALG_WINDOWED_BIG, REF_SQUARE_COMPAT, ALG_CONSTANT_TIME, LADDER_RADIX32, BASE_GENERATOR, GROUP_NO_GATHER  # noqa: F401

__all__ = ["Engine", "DeviceGroup", "shard_range_c", "EcsimdHipError", "CURVES", "P256", "SECP256K1", "lib_path", "load_library",
           "BASE_CLASSICAL", "BASE_MGRY", "OUT_JACOBIAN", "OUT_AFFINE", "ALG_WINDOWED", "ALG_WINDOWED_SIGNED", "ALG_NO_ENDOMORPHISM", "ALG_WINDOWED_BIG", "REF_SQUARE_COMPAT", "ALG_CONSTANT_TIME", "LADDER_RADIX32", "BASE_GENERATOR", "GROUP_NO_GATHER"]
