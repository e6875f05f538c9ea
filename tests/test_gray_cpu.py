"""CPU tests for row f1 of the hot-path scope table, RgbToGrayscale (core/grayscale.go:8-23): the oracle against the
known answers, and the C ABI's argument checks (which run before any device call)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from oracle.np_restatement import np_rgb_to_grayscale
from pigo_amd import core, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gray_golden():
    with open(os.path.join(ROOT, "tests", "golden", "gray_golden.json")) as fh:
        return json.load(fh)["cases"]


def test_reference_test_invariant_uniform_gray_stays_put():
    """core/grayscale_test.go:14-34 builds a uniform (177,177,177,255) image; R=G=B=v, A=255 must give v for every v."""
    v = np.arange(256, dtype=np.uint8)
    img = np.stack([v, v, v, np.full(256, 255, np.uint8)], -1)[None]
    for kind in (oracle.PIX_NRGBA, oracle.PIX_RGBA, oracle.PIX_CANVAS):
        assert (oracle.rgb_to_grayscale(img, kind) == v).all(), kind
    img177 = np.full((10, 10, 4), 177, np.uint8)
    img177[..., 3] = 255
    assert (oracle.rgb_to_grayscale(img177, oracle.PIX_RGBA) == 177).all()


def test_oracle_matches_numpy_restatement_on_every_opaque_colour():
    r, g, b = np.meshgrid(*(np.arange(256, dtype=np.uint8),) * 3, indexing="ij")
    img = np.stack([r, g, b, np.full_like(r, 255)], -1).reshape(4096, 4096, 4)
    for kind in (oracle.PIX_NRGBA, oracle.PIX_CANVAS):
        assert (oracle.rgb_to_grayscale(img, kind) == np_rgb_to_grayscale(img, kind)).all(), kind
    # opaque NRGBA and RGBA read the same 16-bit channels (A = 0xff: c*0x101*0xff/0xff)
    sub = img[::7]
    assert (oracle.rgb_to_grayscale(sub, oracle.PIX_NRGBA) == oracle.rgb_to_grayscale(sub, oracle.PIX_RGBA)).all()


def test_oracle_alpha_premultiplication():
    """color.NRGBA.RGBA(): c*0x101*A/0xff -- every (value, alpha) pair; A=0 gives black."""
    c, a = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    img = np.stack([c, c, c, a], -1)
    o = oracle.rgb_to_grayscale(img, oracle.PIX_NRGBA)
    assert (o == np_rgb_to_grayscale(img, 0)).all()
    assert (o.reshape(256, 256)[:, 0] == 0).all() and (o.reshape(256, 256)[:, 255] == np.arange(256)).all()
    c16 = (c.astype(np.int64) * 257 * a.astype(np.int64)) // 255
    expect = c16 >> 8  # the three weights sum to 1 within rounding: gray ~ floor(c16/256), never more than 1 below
    d = expect.reshape(-1) - o.astype(np.int64)
    assert d.min() >= 0 and d.max() <= 1


def test_oracle_against_gray_goldens_and_strided_rows():
    for rec in gray_golden():
        img = synth.syn_rgba(rec["rows"], rec["cols"], seed=1234, frame_index=rec["frame_index"], opaque_rows=rec["opaque_rows"])
        if rec["rows"] > 500:  # the 1080p case: one kind is enough on CPU
            kinds = ("0",)
        else:
            kinds = ("0", "1", "2")
        for k in kinds:
            o = oracle.rgb_to_grayscale(img, int(k))
            assert hashlib.sha256(o.tobytes()).hexdigest() == rec["kinds"][k]["sha256"], (rec["name"], k)
            assert [int(v) for v in o[:16]] == rec["kinds"][k]["head"]
    big = synth.syn_rgba(50, 80, frame_index=9, opaque_rows=10)
    view = big[5:45, 8:72]  # a Go sub-image: same Pix, larger Stride
    assert (oracle.rgb_to_grayscale(view, 0) == oracle.rgb_to_grayscale(np.ascontiguousarray(view), 0)).all()


def test_abi_argument_checks_need_no_gpu():
    img = synth.syn_rgba(6, 5)
    with pytest.raises(ValueError):
        core.RgbToGrayscale(img, kind=7)
    with pytest.raises(ValueError):
        core.RgbToGrayscale(np.zeros((4, 4, 3), np.uint8))
    assert core.RgbToGrayscale(np.zeros((0, 9, 4), np.uint8)).size == 0  # make([]uint8, 0)
    L = core.load_library()
    out = np.zeros(30, np.uint8)
    flat = np.ascontiguousarray(img).ravel()
    # Pix too short for (width, height, stride): src.At(x, y) would index past it -> the reference panics (grayscale.go:14)
    st = L.pigo_rgb_to_grayscale(0, flat.ctypes.data, flat.size - 1, 5, 6, 20, 0, out.ctypes.data, out.size)
    assert st == core.ERR_PANIC
    with pytest.raises(core.PigoPanic):
        core.check(st)
    assert L.pigo_rgb_to_grayscale(0, flat.ctypes.data, flat.size, 5, 6, 19, 0, out.ctypes.data, out.size) == core.ERR_PARAM
    assert L.pigo_rgb_to_grayscale(0, flat.ctypes.data, flat.size, 5, 6, 20, 0, out.ctypes.data, 29) == core.ERR_CAPACITY
    assert L.pigo_gray_batch(0, None, 0, 20, 5, 6, 0, 0, None, 0, 5, None) == core.PIGO_OK   # zero frames: nothing to do
    assert L.pigo_gray_batch(0, None, 0, 20, 5, 6, 3, 1, None, 0, 5, None) == core.ERR_PARAM  # unknown kind
    assert L.pigo_gray_batch(0, None, 0, 20, 5, 6, 0, 1, None, 0, 5, None) == core.ERR_PARAM  # NULL pointers
    if L.pigo_device_count() == 0:  # no GPU: must fail with a HIP error, never fall back to a CPU path
        with pytest.raises(core.PigoError):
            core.RgbToGrayscale(img)
