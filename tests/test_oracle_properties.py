"""Property tests that cross-pin the two independent CPU restatements (oracle/pigo_oracle.c vs oracle/np_restatement.py)
on generated inputs: neither can run against Go here (PARITY UNPINNED), so every place where the C text and the NumPy
text were written separately from the reference must at least agree with each other everywhere we can reach."""
import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle
from oracle.np_restatement import NpPuploc, np_rgb_to_grayscale
from pigo_amd import synth

SETTINGS = dict(derandomize=True, deadline=None, suppress_health_check=list(HealthCheck))

_PUP = {}


def _pup(name):
    if name not in _PUP:
        pk = synth.cascade_bytes(name)
        _PUP[name] = (oracle.OraclePuploc.unpack(pk), NpPuploc(pk))
    return _PUP[name]


@settings(max_examples=60, **SETTINGS)
@given(h=st.integers(1, 9), w=st.integers(1, 9), kind=st.integers(0, 2), seed=st.integers(0, 2**31 - 1),
       extreme=st.sampled_from([None, 0, 1, 127, 128, 254, 255]))
def test_gray_c_equals_numpy(h, w, kind, seed, extreme):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    if extreme is not None:
        img[rng.random((h, w, 4)) < 0.5] = extreme
    assert (oracle.rgb_to_grayscale(img, kind) == np_rgb_to_grayscale(img, kind)).all()


@settings(max_examples=40, **SETTINGS)
@given(name=st.sampled_from(["puploc", "lps/lp38", "lps/lp81"]), rows=st.integers(1, 60), cols=st.integers(1, 60),
       r=st.floats(-50, 110, width=32), c=st.floats(-50, 110, width=32), s=st.floats(0.0, 300.0, width=32),
       flip=st.booleans(), angle=st.sampled_from([None, 0.001, 0.03125, 0.3, 0.5, 0.99, 1.0]), seed=st.integers(0, 2**31 - 1))
def test_puploc_classify_c_equals_numpy(name, rows, cols, r, c, s, flip, angle, seed):
    """One perturbation through classifyRegion / classifyRotatedRegion: tiny and degenerate images (1 x 1 included), starting
    points outside the image, scale 0, every clamp and the int8 negation of flipV."""
    o, n = _pup(name)
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    a = o.classify(r, c, s, img, rows, cols, cols, angle=angle or 0.0, rotated=angle is not None, flip_v=flip)
    b = n.classify([r], [c], [s], img, rows, cols, cols, flip_v=flip, angle=angle)
    assert a[0] == b[0][0] and a[1] == b[1][0] and a[2] == b[2][0], (a, b)


@settings(max_examples=25, **SETTINGS)
@given(perturbs=st.integers(0, 63), row=st.integers(-10, 250), col=st.integers(-10, 330), scale=st.floats(1.0, 120.0, width=32),
       flip=st.booleans(), rot=st.booleans(), seed=st.integers(0, 2**20), reuse=st.booleans())
def test_run_detector_c_equals_numpy_with_pool_state(perturbs, row, col, scale, flip, rot, seed, reuse):
    o, n = _pup("puploc")
    img = synth.syn_noise(240, 320, seed=7, frame_index=seed % 5)
    rnd = synth.syn_uniform32(189, seed=seed, index=1)
    po, pn = np.zeros((3, 63), np.float32), np.zeros((3, 63), np.float32)
    if reuse:  # a used pool object: stale entries from an earlier call
        po[:] = pn[:] = np.sort(synth.syn_uniform32(189, seed=seed, index=2).reshape(3, 63) * 300, axis=1)
    angle = 0.4 if rot else 0.0
    a = o.run_detector(row, col, float(scale), perturbs, img, 240, 320, 320, angle, flip, rnd, po)
    b = n.run_detector(row, col, float(scale), perturbs, img, 240, 320, 320, angle, flip, rnd, pn)
    assert a == b and (po == pn).all()
