"""Guards on the SHIPPED gfx950 ISA of every translation unit (CPU suite; hipcc cross-compiles here).

The Makefile keeps each unit's device assembly beside its object (-save-temps=obj), so these checks read what was really linked into
ecsimd_amd/libecsimd_hip.so -- no second compilation.  tools/asm_clobber_check.py: no inline-asm Comba column may have had a
multiplicand's register reused for one of its own results (a compiler defect met in round 4: an early-clobber output was given the
register of a live input and a diagnostic kernel returned a wrong Z; see the tool's header)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_clobber_check  # noqa: E402


@pytest.fixture(scope="module")
def listings():
    subprocess.run(["make", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(ROOT, "ecsimd_amd", "csrc"), "ARCH=gfx950"], check=True, capture_output=True)
    files = sorted(glob.glob(os.path.join(ROOT, "build", "csrc", "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    assert len(files) >= 17, "the Makefile no longer leaves the device listings in build/csrc (-save-temps=obj)"
    return files


@pytest.mark.timeout(1200)
def test_no_asm_block_overwrites_its_own_multiplicand(listings):
    bad = [b for f in listings for b in asm_clobber_check.check(f)]
    assert not bad, bad[:5]


def test_the_checker_refuses_the_defect_it_was_written_for(tmp_path):
    """The block the compiler emitted for column 13 of z = z * zz in round 4's first k_zdau_repeat<32>: v3 is z.w[7], a multiplicand of the
    second product and of the next column, and was handed to the carry counter."""
    planted = tmp_path / "planted.s"
    planted.write_text("k_planted:\n\t;;#ASMSTART\n\tv_mad_u64_u32 v[48:49], vcc, v2, v17, v[48:49]\n\tv_addc_co_u32 v3, vcc, 0, 0, vcc\n"
                       "\tv_mad_u64_u32 v[48:49], vcc, v3, v19, v[48:49]\n\tv_addc_co_u32 v3, vcc, 0, v3, vcc\n\t;;#ASMEND\n")
    assert len(asm_clobber_check.check(str(planted))) == 1
    clean = tmp_path / "clean.s"
    clean.write_text("k_clean:\n\t;;#ASMSTART\n\tv_mad_u64_u32 v[48:49], vcc, v2, v17, v[48:49]\n\tv_addc_co_u32 v5, vcc, 0, 0, vcc\n"
                     "\tv_mad_u64_u32 v[48:49], vcc, v3, v19, v[48:49]\n\tv_addc_co_u32 v5, vcc, 0, v5, vcc\n\t;;#ASMEND\n")
    assert asm_clobber_check.check(str(clean)) == []
