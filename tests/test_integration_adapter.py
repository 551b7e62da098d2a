"""The reference-side adapter of INTEGRATION.md section 2 (integration/scalar_mult_p256_adapter.cpp).

CPU part (build container only: skipped where /root/reference does not exist): the adapter compiles against the REFERENCE'S OWN headers and links
against libecsimd_hip.so, as a shared object a maintainer would ship and as the test binary oracle/_ref/adapter_driver (oracle/Makefile).
GPU part: that binary -- built here, shipped to the GPU box like oracle/_ref/libecsimd_ref.so -- EXECUTES the adapter beside the real reference in
one process: the ScalarMult scenarios of the reference's tests/curve_group.cpp:117-173, lane-distinct wides compared limb for limb with
curve_group<curve_nist_p256>::scalar_mult (ECSIMD_HIP_REF_SQUARE_COMPAT on: not one limb may differ), the four-lane and the batch form."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SRC = os.path.join(ROOT, "integration", "scalar_mult_p256_adapter.cpp")
OUT = os.path.join(ROOT, "build", "tests", "libscalar_mult_p256_adapter.so")
DRIVER = os.path.join(ROOT, "oracle", "_ref", "adapter_driver")


def have_reference():
    return os.path.isdir(os.path.join(REF, "include", "ecsimd"))


def build_driver():
    """oracle/_ref/adapter_driver, rebuilt where the reference's sources are (make decides whether it is stale); elsewhere the shipped file as it is."""
    if have_reference():
        import ecsimd_amd
        if not os.path.exists(ecsimd_amd.lib_path()):
            import __graft_entry__
            __graft_entry__.build()
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/adapter_driver"], check=True, capture_output=True)
    return DRIVER if os.path.exists(DRIVER) else None


@pytest.mark.timeout(600)
def test_reference_side_adapter_compiles_and_links():
    if not have_reference():
        pytest.skip("the reference's sources are not on this machine")
    import ecsimd_amd
    if not os.path.exists(ecsimd_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    libdir = os.path.join(ROOT, "ecsimd_amd")
    newest = max(os.path.getmtime(p) for p in (SRC, SRC[:-4] + ".h", os.path.join(ROOT, "include", "ecsimd_hip.h")))
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        # the reference needs g++ (not clang) and plain AVX2; its include directory comes FIRST so that
        # <ecsimd/...> is the reference's, and only <ecsimd_hip.h> comes from this repo
        subprocess.run(["g++", "-std=c++20", "-O1", "-mavx2", "-fPIC", "-shared", "-I", os.path.join(REF, "include"), "-I", os.path.join(REF, "third-party"),
                        "-I", os.path.join(ROOT, "include"), SRC, "-o", OUT, "-L", libdir, "-lecsimd_hip", "-Wl,-rpath," + libdir, "-Wl,--no-undefined"], check=True)
    syms = subprocess.run(["nm", "-DC", "--defined-only", OUT], capture_output=True, text=True, check=True).stdout
    assert "scalar_mult_p256(eve::" in syms and "scalar_mult_p256(std::span<" in syms, "the adapter does not export the reference's entry point and its batch form"
    undefined = subprocess.run(["nm", "-D", "--undefined-only", OUT], capture_output=True, text=True, check=True).stdout
    for f in ("ecsimd_hip_init", "ecsimd_hip_scalar_mult_p256", "ecsimd_hip_memcpy_h2d", "ecsimd_hip_memcpy_d2h", "ecsimd_hip_malloc", "ecsimd_hip_free",
              "ecsimd_hip_wide4_to_lanes", "ecsimd_hip_lanes_to_wide4"):
        assert f in undefined, f
    assert "libecsimd_hip.so" in subprocess.run(["ldd", OUT], capture_output=True, text=True).stdout
    exe = build_driver()
    assert exe and "libecsimd_hip.so" in subprocess.run(["ldd", exe], capture_output=True, text=True).stdout


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_the_adapter_executes_beside_the_reference_on_the_gpu():
    exe = build_driver()
    if exe is None:
        pytest.skip("oracle/_ref/adapter_driver did not travel here and cannot be built without the reference's sources")
    r = subprocess.run([exe, "512", str(1 << 15), str((1 << 18) + 1234)], capture_output=True, text=True, timeout=800)     # the third size: three chunks on two contexts, a ragged tail
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0 and "adapter_driver ok (0 failed)" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert "scenarios of tests/curve_group.cpp ScalarMult through the adapter: ok" in r.stdout
    m = re.search(r"lane-distinct: (\d+) lanes .*?: (\d+) with REF_SQUARE_COMPAT \(must be 0\), (\d+) with exact squaring .*?: (\d+) differing", r.stdout)
    assert m and int(m.group(1)) == 2048 and int(m.group(2)) == 0 and int(m.group(4)) == 0
    assert re.search(r"batch form: 32768 wides = 131072 lanes in one call", r.stdout)
    assert re.search(r"batch form: 263378 wides = 1053512 lanes in one call", r.stdout) and re.search(r"the same batch with REF_SQUARE_COMPAT: 64 sampled wides spread over the batch, 0 lanes differ", r.stdout)
    assert "lane transposition: on the device" in r.stdout                       # the spans travelled as raw bytes (the adapter's layout check passed)
    # ... and the adapter's other path (the per-lane conversion on the host, taken when a compiler lays the reference's types out differently): the same checks
    r = subprocess.run([exe, "128", "0"], capture_output=True, text=True, timeout=800, env=dict(os.environ, ECSIMD_ADAPTER_HOST_TRANSPOSE="1"))
    assert r.returncode == 0 and "adapter_driver ok (0 failed)" in r.stdout and "lane transposition: per lane on the host" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
