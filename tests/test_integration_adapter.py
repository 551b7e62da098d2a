"""CPU suite: the reference-side adapter of INTEGRATION.md section 2 (integration/scalar_mult_p256_adapter.cpp) really
compiles against the REFERENCE'S OWN headers and links against libecsimd_hip.so.  Build container only: skipped where
/root/reference does not exist (the GPU box).  Nothing is executed -- the adapter needs a GPU; this is the
"a maintainer can add this file" check."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SRC = os.path.join(ROOT, "integration", "scalar_mult_p256_adapter.cpp")
OUT = os.path.join(ROOT, "build", "tests", "libscalar_mult_p256_adapter.so")


@pytest.mark.timeout(600)
def test_reference_side_adapter_compiles_and_links():
    if not os.path.isdir(os.path.join(REF, "include", "ecsimd")):
        pytest.skip("the reference's sources are not on this machine")
    import ecsimd_amd
    if not os.path.exists(ecsimd_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    libdir = os.path.join(ROOT, "ecsimd_amd")
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "ecsimd_hip.h"))):
        # the reference needs g++ (not clang) and plain AVX2; its include directory comes FIRST so that
        # <ecsimd/...> is the reference's, and only <ecsimd_hip.h> comes from this repo
        subprocess.run(["g++", "-std=c++20", "-O1", "-mavx2", "-fPIC", "-shared", "-I", os.path.join(REF, "include"), "-I", os.path.join(REF, "third-party"),
                        "-I", os.path.join(ROOT, "include"), SRC, "-o", OUT, "-L", libdir, "-lecsimd_hip", "-Wl,-rpath," + libdir, "-Wl,--no-undefined"], check=True)
    syms = subprocess.run(["nm", "-DC", "--defined-only", OUT], capture_output=True, text=True, check=True).stdout
    assert "scalar_mult_p256(" in syms, "the adapter does not export the reference's entry point"
    undefined = subprocess.run(["nm", "-D", "--undefined-only", OUT], capture_output=True, text=True, check=True).stdout
    for f in ("ecsimd_hip_init", "ecsimd_hip_scalar_mult_p256", "ecsimd_hip_memcpy_h2d", "ecsimd_hip_memcpy_d2h", "ecsimd_hip_malloc", "ecsimd_hip_free"):
        assert f in undefined, f
    assert "libecsimd_hip.so" in subprocess.run(["ldd", OUT], capture_output=True, text=True).stdout
