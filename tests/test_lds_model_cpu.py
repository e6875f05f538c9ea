"""The LDS bank model behind profiles/r05_region_model.md (scripts/lds_conflict_sim.py: two groups of 32 lanes, one LDS cycle per
group plus one per extra distinct dword on its busiest bank, bank = (byte address / 4) mod 32) on the patterns whose cost the
microbenchmark scripts/micro/lds_gather.hip MEASURED on an MI355X (profiles/r05_lds_gather.txt): the model must give the same
numbers the microbenchmark printed next to its measurements."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_bank_model_on_the_microbenchmark_patterns():
    from lds_conflict_sim import group_cycles
    lane = np.arange(64, dtype=np.int64)
    one = lambda a: group_cycles(np.asarray(a, dtype=np.int64).reshape(1, 64))  # noqa: E731
    assert one(lane * 4) == 2.0                      # conflict-free: measured 2.60 (2.0 array cycles + issue)
    assert one(np.full(64, 1234)) == 2.0             # one address: broadcast
    for step in (1, 2, 3):
        assert one(5000 + lane * step) == 2.0        # neighbouring windows at the same node
    assert one(5000 + lane * 5) == 4.0               # 32 lanes span 160 B > 32 banks: measured 4.15
    assert one(30 * 332 + 40 + lane * 332) == 2.0    # one column, 64 rows of an odd-dword pitch
    assert one(np.where(lane < 32, lane * 4, -1)) == 2.0  # an idle group still takes its cycle
    # every lane its own dword on ONE bank: 32 distinct dwords per group
    assert one(lane * 128) == 64.0
    # random addresses: 7.1 +- 0.2 (measured 7.16)
    rng = np.random.default_rng(5)
    v = group_cycles(rng.integers(0, 128 << 10, size=(4000, 64)))
    assert 6.9 < v < 7.4, v


def test_the_committed_microbenchmark_output_agrees_with_the_model_within_5_percent():
    """profiles/r05_lds_gather.txt: every CONFLICTING pattern's measured cycles within 5 % of the 32-bank model printed beside it."""
    import re
    rows = 0
    with open(os.path.join(ROOT, "profiles", "r05_lds_gather.txt")) as fh:
        for line in fh:
            m = re.match(r"^(u8|u16|b64)\s+(.+?)\s+[\d.]+ ms\s+([\d.]+) cycles .* model:\s+([\d.]+) \(32 banks\)", line)
            if not m or float(m.group(4)) < 3.9:
                continue
            rows += 1
            meas, model = float(m.group(3)), float(m.group(4))
            assert abs(meas - model) / model < 0.05, line
    assert rows >= 15, rows


def test_the_microbenchmarks_compile_for_gfx950(tmp_path):
    """scripts/micro/*.hip are evidence (profiles/r03_ta_gather.txt, r05_lds_gather.txt): they must keep building with the image's hipcc."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    for name in ("lds_gather", "ta_gather"):
        src = os.path.join(ROOT, "scripts", "micro", name + ".hip")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", src, "-o", str(tmp_path / name)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
