"""CPU check of the arithmetic identity behind k_scan_region's window decode (reg_split): the truncated float32 product
(f + 0.5) * (1 / nj) is floor(f / nj) for every window index and row length the region kernel can see, so the decode needs no
correction step.  Compiles and runs tests/region_decode_check.c (all 2^32 pairs, reciprocal perturbed by -3 / 0 / +3 ulp)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_truncated_float_quotient_is_exact(tmp_path):
    exe = str(tmp_path / "region_decode_check")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", os.path.join(HERE, "region_decode_check.c"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches: 0" in out.stdout
