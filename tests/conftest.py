import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_oracle():
    from oracle import loader
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libecsimd_oracle.so"], check=True)   # gcc is everywhere
    return loader


@pytest.fixture(scope="session")
def oracle():
    """The C restatement (test infrastructure) in EXACT mode: square(a) == mul(a, a), the arithmetic
    the reference specifies.  This is what the HIP path is checked against."""
    return _build_oracle().Oracle(faithful=False)


@pytest.fixture(scope="session")
def oracle_faithful():
    """The same restatement, bug-for-bug: reproduces the reference's dropped carry in square()
    (oracle/ecsimd_oracle.c bn_square).  Used to pin the restatement against the compiled reference."""
    return _build_oracle().Oracle(faithful=True)


@pytest.fixture(scope="session")
def reference():
    """The real reference behind oracle/_ref (only where it was built; never required)."""
    from oracle import loader
    if not loader.reference_available():
        pytest.skip("oracle/_ref/libecsimd_ref.so not present (built only where /root/reference exists)")
    return loader.Reference()


@pytest.fixture(scope="session")
def openssl():
    """OpenSSL's libcrypto behind oracle/ossl_check.c: an implementation independent of the reference and of
    the restatement (level-A cross-check).  Skipped where the headers are missing."""
    import subprocess
    from oracle import loader
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ossl"], check=True)
    if not loader.openssl_available():
        pytest.skip("OpenSSL development headers not installed")
    return loader.OpenSSLCheck()


@pytest.fixture(scope="session")
def engine():
    """The product: HIP kernels through the C ABI.  No fallback: fails if the library or GPU is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (they are marked @pytest.mark.gpu)"
    from ecsimd_amd import Engine
    return Engine(0)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_moduli():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_moduli_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_curves():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_curves_vectors.json")) as f:
        return json.load(f)
