"""Register and scratch budget of the region kernel, checked at compile time (hipcc cross-compiles gfx950 without a GPU).

k_scan_region shares every CU with the side chain's workgroups: four region waves per SIMD at 96 VGPRs leave 128 of
the SIMD's 512 registers to the chain's waves.  Round 6 measured both ways of breaking that silently: a variant that needed 98
VGPRs (104 allocated) ran the same alone and 5.62 instead of 5.22 ms next to the chain, and at a cap of 96 the kernel used to
spill to scratch, i.e. through the vector-memory path the chain saturates (profiles/r06_experiments.md).  The compiler's own
report is the check: VGPRs <= 96 and no scratch for both instantiations (upright: classifyRegion, core/pigo.go:113-147; rotated:
classifyRotatedRegion, core/pigo.go:150-191)."""
import os
import re
import subprocess

from pigo_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_region_kernel_fits_next_to_the_side_chain(tmp_path):
    out = tmp_path / "libpigo_hip_resources.so"
    cmd = [build.hipcc()] + list(build.HIPCC_FLAGS) + ["-Rpass-analysis=kernel-resource-usage"] + build.SOURCES + ["-o", str(out)]
    r = subprocess.run(cmd, cwd=build.CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = r.stderr
    seen = 0
    for m in re.finditer(r"Function Name: (\S*k_scan_region\S*)(.*?)(?=Function Name:|\Z)", text, re.S):
        body = m.group(2)
        vgprs = int(re.search(r"VGPRs: (\d+)", body).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", body).group(1))
        vspill = re.search(r"VGPRs Spill: (\d+)", body)
        assert vgprs <= 96, f"{m.group(1)} needs {vgprs} VGPRs: more than 96 loses the side chain's co-residency (~0.4 ms per step)"
        assert scratch == 0 and (vspill is None or int(vspill.group(1)) == 0), f"{m.group(1)}: scratch {scratch} B/lane -- spills go through the vector-memory path the side chain saturates"
        seen += 1
    assert seen == 2, f"expected the upright and the rotated instantiation of k_scan_region in the report, found {seen}"
