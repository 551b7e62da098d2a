"""The C++ host API (include/ecsimd/*.h, the reference's names over the C ABI).

CPU part: the headers compile under g++ AND ROCm clang (the reference's own headers do not
compile with clang -- SURVEY.md 8(c)) and link against libecsimd_hip.so.
GPU part: tests/cpp/host_api_tests.cpp re-runs every known-answer scenario of the reference's
gtest files through those headers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_api_tests.cpp")
OUT = os.path.join(ROOT, "build", "tests", "host_api_tests")
INC = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp")]


def build_binary():
    import ecsimd_amd
    if not os.path.exists(ecsimd_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    libdir = os.path.join(ROOT, "ecsimd_amd")
    newest_src = max(os.path.getmtime(p) for p in [SRC, os.path.join(ROOT, "tests", "cpp", "mini_test.h")] +
                     [os.path.join(ROOT, "include", "ecsimd", f) for f in os.listdir(os.path.join(ROOT, "include", "ecsimd"))])
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < newest_src:
        subprocess.run(["g++", "-std=c++20", "-O1", "-Wall", "-Werror", *INC, SRC, "-o", OUT, "-L", libdir, "-lecsimd_hip",
                        "-Wl,-rpath," + libdir], check=True)
    return OUT


BENCH_SRC = os.path.join(ROOT, "benchs", "host_benchs.cpp")
BENCH_OUT = os.path.join(ROOT, "build", "tests", "host_benchs")


def build_benchs():
    """The reference's benchs/curve_group.cpp and benchs/ops.cpp, re-hosted on the C++ host API."""
    build_binary()                                      # makes sure the library exists
    libdir = os.path.join(ROOT, "ecsimd_amd")
    if not os.path.exists(BENCH_OUT) or os.path.getmtime(BENCH_OUT) < max(os.path.getmtime(BENCH_SRC), os.path.getmtime(OUT)):
        subprocess.run(["g++", "-std=c++20", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), BENCH_SRC, "-o", BENCH_OUT,
                        "-L", libdir, "-lecsimd_hip", "-Wl,-rpath," + libdir], check=True)
    return BENCH_OUT


def test_rehosted_benchs_compile():
    assert os.path.exists(build_benchs())


@pytest.mark.gpu
def test_rehosted_benchs_run():
    r = subprocess.run([build_benchs(), "14", "2"], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for name in ("scalar_mult_p256 + to_affine", "scalar_mult_p256_1s + to_affine", "add_256", "mul_256", "sqr_256", "mgry_sqr_256", "mgry_reduce_512"):
        assert name in r.stdout, name


def test_headers_compile_and_link_with_gxx():
    build_binary()
    assert os.path.exists(OUT)


def test_headers_compile_with_rocm_clang():
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        clang = shutil.which("clang++")
    if not clang:
        pytest.skip("no clang++")
    subprocess.run([clang, "-std=c++20", "-fsyntax-only", "-Wall", *INC, SRC], check=True)


@pytest.mark.gpu
def test_reference_scenarios_through_the_cpp_api():
    exe = build_binary()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " 0 failed" in r.stdout
