import json
import os
import sys

import numpy as np
import pytest

# the parity suite forces schedule variants, queue sizes and code paths through the library's tuning switches: those are only
# honoured with PIGO_TUNING=1 (pigo_hip.hip: env_int)
# (PIGO_TEST_NO_TUNING=1 -- tests/test_gpu_production_env.py -- runs the suite the way production runs the library: switches inert)
if os.environ.get("PIGO_TEST_NO_TUNING") == "1":
    os.environ.pop("PIGO_TUNING", None)
else:
    os.environ.setdefault("PIGO_TUNING", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """A fresh checkout has no libpigo_hip.so / libpigo_oracle.so (built artefacts are not in git): compile them once
    (hipcc cross-compiles gfx950 without a GPU).  Building is not falling back: without the library every product call
    still fails loudly."""
    from pigo_amd import build as hip_build
    import oracle
    hip_build.build()
    oracle.build()


@pytest.fixture(scope="session")
def packet():
    from pigo_amd import synth
    return synth.facefinder_bytes()


@pytest.fixture(scope="session")
def gray():
    from pigo_amd import synth
    return synth.sample_gray()


@pytest.fixture(scope="session")
def orc(packet):
    """The CPU oracle (test infrastructure)."""
    import oracle
    return oracle.OraclePigo.unpack(packet)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden_cases.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def pg(packet):
    """The product: Pigo handle on cuda:0 through the C ABI (GPU tests only)."""
    from pigo_amd import core
    return core.NewPigo(0).Unpack(packet)


def golden_image(name):
    """Rebuild the input image of a golden case (same recipes as tests/golden/make_golden.py)."""
    from pigo_amd import synth
    g = synth.sample_gray()
    if name.startswith("sample_"):
        return g
    if name.startswith("land_"):
        return np.ascontiguousarray(g[40:340, :])
    if name.startswith("faces_480x640"):
        return synth.syn_faces(480, 640, seed=1234, frame_index=0)
    if name == "noise_240x320":
        return synth.syn_noise(240, 320, seed=1234, frame_index=3)
    if name == "faces_200x260_dim300":
        s = np.zeros((200, 300), dtype=np.uint8)
        s[:, :260] = synth.syn_faces(200, 260, seed=7, frame_index=1)
        s[:, 260:] = 255
        return s
    raise KeyError(name)


def f32_from_hex(h):
    return np.frombuffer(bytes.fromhex(h), dtype="<f4")[0]


def assert_same_dets(got, want, what="", q_tol=0.0):
    """Integer triples bit-exact; q within q_tol (0.0 = bit-exact)."""
    assert len(got) == len(want), f"{what}: {len(got)} detections, want {len(want)}"
    for i, (a, b) in enumerate(zip(got, want)):
        ta = (int(a["row"]), int(a["col"]), int(a["scale"]))
        tb = (int(b["row"]), int(b["col"]), int(b["scale"]))
        assert ta == tb, f"{what}: detection {i}: {ta} != {tb}"
        qa, qb = np.float32(a["q"]), np.float32(b["q"])
        if q_tol == 0.0:
            assert qa == qb, f"{what}: detection {i} {ta}: q {qa!r} != {qb!r}"
        else:
            assert abs(float(qa) - float(qb)) <= q_tol, f"{what}: detection {i} {ta}: q {qa!r} vs {qb!r}"
