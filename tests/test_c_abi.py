"""The C ABI from plain C99 (tests/c/abi_smoke.c): compiles with gcc -std=c99 -pedantic -Werror against include/ecsimd_hip.h alone and links
with -lecsimd_hip (CPU part); on a GPU it runs one context and a three-member device group over the same batch (GPU part)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
OUT = os.path.join(ROOT, "build", "tests", "abi_smoke")


def build():
    import ecsimd_amd
    if not os.path.exists(ecsimd_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    libdir = os.path.join(ROOT, "ecsimd_amd")
    newest = max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "ecsimd_hip.h")))
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", OUT,
                        "-L", libdir, "-lecsimd_hip", "-Wl,-rpath," + libdir], check=True)
    return OUT


def test_c99_caller_compiles_and_links():
    assert os.path.exists(build())


@pytest.mark.gpu
def test_c99_caller_runs():
    r = subprocess.run([build()], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "abi_smoke ok" in r.stdout, r.stdout[-1000:] + r.stderr[-1000:]
