"""GPU parity tests for RgbToGrayscale (core/grayscale.go:8-23): the HIP kernels through the
C ABI against the CPU oracle and the committed known answers.  Byte output: the bar is bit-exact."""
import hashlib

import numpy as np
import pytest

import oracle
from pigo_amd import core, synth

from test_gray_cpu import gray_golden

pytestmark = pytest.mark.gpu

KINDS = (core.PIX_NRGBA, core.PIX_RGBA, core.PIX_CANVAS)


def test_gray_goldens_all_kinds():
    for rec in gray_golden():
        img = synth.syn_rgba(rec["rows"], rec["cols"], seed=1234, frame_index=rec["frame_index"], opaque_rows=rec["opaque_rows"])
        for k in KINDS:
            g = core.RgbToGrayscale(img, kind=k)
            assert g.shape == (rec["rows"] * rec["cols"],)
            assert hashlib.sha256(g.tobytes()).hexdigest() == rec["kinds"][str(k)]["sha256"], (rec["name"], k)


def test_gray_every_opaque_colour_and_every_alpha_pair():
    r, g, b = np.meshgrid(*(np.arange(256, dtype=np.uint8),) * 3, indexing="ij")
    img = np.stack([r, g, b, np.full_like(r, 255)], -1).reshape(4096, 4096, 4)   # dense -> the 16-byte-per-lane kernel
    for k in (core.PIX_NRGBA, core.PIX_CANVAS):
        assert (core.RgbToGrayscale(img, kind=k) == oracle.rgb_to_grayscale(img, k)).all(), k
    c, a = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for chans in ((c, c, c), (c, 255 - c, c // 2), (a, c, 255 - a)):
        pa = np.ascontiguousarray(np.stack([*chans, a], -1))
        for k in KINDS:
            assert (core.RgbToGrayscale(pa, kind=k) == oracle.rgb_to_grayscale(pa, k)).all(), k


def test_gray_reference_test_invariant():
    """core/grayscale_test.go:14-34: the uniform (177,177,177,255) *image.RGBA stays 177."""
    img = np.full((10, 10, 4), 177, np.uint8)
    img[..., 3] = 255
    assert (core.RgbToGrayscale(img, kind=core.PIX_RGBA) == 177).all()


def test_gray_ragged_shapes_and_strided_rows():
    big = synth.syn_rgba(70, 90, frame_index=5, opaque_rows=20)
    for view in (big[:1, :1], big[:1, :7], big[:3, :64], big[:33, :65], big[5:64, 9:72], big[:, :88], big):
        for k in KINDS:
            got = core.RgbToGrayscale(view, kind=k)
            assert (got == oracle.rgb_to_grayscale(view, k)).all(), (view.shape, k)
    assert core.RgbToGrayscale(np.zeros((0, 5, 4), np.uint8)).size == 0
    with pytest.raises(ValueError):
        core.RgbToGrayscale(big, kind=9)


def test_gray_batch_feeds_the_scan(pg, orc):
    """Device-resident pipeline: RGBA frames -> pigo_gray_batch -> pigo_plan_run, no host pass in between."""
    import torch
    from pigo_amd import batch
    rows, cols, n = 240, 320, 5
    gray_ref = synth.make_frames("faces", n, rows, cols, seed=1234)
    rgba = np.zeros((n, rows, cols, 4), np.uint8)
    noise = synth.syn_rgba(rows, cols, frame_index=11, opaque_rows=rows // 2)
    for f in range(n):
        if f % 2 == 0:  # R=G=B=gray, opaque: the conversion must give the frame back
            rgba[f, ..., :3] = gray_ref[f][..., None]
            rgba[f, ..., 3] = 255
        else:
            rgba[f] = np.roll(noise, f, axis=1)
    d_rgba = torch.from_numpy(rgba).cuda()
    for k in KINDS:
        g = batch.rgb_to_grayscale(d_rgba, kind=k)
        torch.cuda.synchronize()
        gh = g.cpu().numpy()
        for f in range(n):
            assert (gh[f].ravel() == oracle.rgb_to_grayscale(rgba[f], k)).all(), (k, f)
    # row pitch wider than the image (ImageParams.Dim > Cols): padding bytes untouched, generic kernel
    dim = cols + 24
    out = torch.full((n, rows, dim), 7, dtype=torch.uint8, device="cuda")
    batch.rgb_to_grayscale(d_rgba, kind=core.PIX_NRGBA, out=out, dim=dim)
    torch.cuda.synchronize()
    oh = out.cpu().numpy()
    assert (oh[:, :, cols:] == 7).all()
    for f in range(n):
        assert (oh[f, :, :cols].ravel() == oracle.rgb_to_grayscale(rgba[f], 0)).all()
    # and straight into the scan
    g = batch.rgb_to_grayscale(d_rgba, kind=core.PIX_NRGBA)
    plan = batch.ScanPlan(pg, rows, cols, MinSize=20, MaxSize=200, ShiftFactor=0.1, ScaleFactor=1.1, max_frames=n, det_cap=4096)
    dets, counts = plan.alloc_outputs(n)
    plan.run(g, dets, counts)
    torch.cuda.synchronize()
    plan.status()
    lists = batch.dets_to_numpy(dets, counts)
    for f in (0, 2):
        want = orc.run_cascade(gray_ref[f], rows, cols, cols, 20, 200, 0.1, 1.1)
        assert len(lists[f]) == len(want) > 0
        assert (lists[f]["row"] == want["row"]).all() and (lists[f]["col"] == want["col"]).all()
        assert (lists[f]["q"] == want["q"]).all()
