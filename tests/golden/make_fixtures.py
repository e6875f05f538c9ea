#!/usr/bin/env python3
"""Regenerate the input fixtures that cannot travel to the GPU box.

Run in the BUILD container only (needs /root/reference, which is absent on the GPU box):

    python tests/golden/make_fixtures.py

Writes
  pigo_amd/data/facefinder               the reference's face cascade (model DATA, not source;
                                         /root/reference/cascade/facefinder, 239,632 B)
  pigo_amd/data/puploc, lps/lp*          the pupil / facial-landmark cascades (model DATA; cascade/puploc 1,228,416 B,
                                         the nine cascade/lps/lp* files 736,816 B each) for scope rows f2/f3
  pigo_amd/data/sample_gray_320x400.bin  testdata/sample.jpg decoded with Pillow and converted with
                                         the reference's gray formula (core/grayscale.go:8-23):
                                         uint8((0.299*r16 + 0.587*g16 + 0.114*b16) / 256) in float64
                                         with r16 = R*257 (color.RGBA() of an opaque 8-bit pixel).

NOTE: Go's image/jpeg decoder differs from libjpeg-turbo (Pillow) by a few LSBs, so
this gray buffer is NOT bit-identical to what the Go test-suite feeds RunCascade.  Parity on
"sample.jpg" is therefore defined on this committed gray fixture, not on the JPEG.
"""
import hashlib
import os
import shutil
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "..", "..", "pigo_amd", "data")

EXPECT = {
    "cascade/facefinder": "d8014993e7298c7b1865d1f8b855d6dbf4ec5c808bf879e2091ab6837abf90cd",
    "testdata/sample.jpg": "09ee4f7085e1eee6f48d3a2b11791c4c008e2dc0c6b2b2462067104e3030915e",
}
EXTRA_CASCADES = ["cascade/puploc"] + ["cascade/lps/" + n for n in ("lp312", "lp38", "lp42", "lp44", "lp46", "lp81", "lp82", "lp84", "lp93")]


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    for rel, want in EXPECT.items():
        got = sha(os.path.join(REF, rel))
        if got != want:
            sys.exit(f"{rel}: sha256 {got} != {want}")
    os.makedirs(DATA, exist_ok=True)
    shutil.copyfile(os.path.join(REF, "cascade/facefinder"), os.path.join(DATA, "facefinder"))
    os.chmod(os.path.join(DATA, "facefinder"), 0o644)
    for rel in EXTRA_CASCADES:
        dst = os.path.join(DATA, rel[len("cascade/"):])
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
        os.chmod(dst, 0o644)
        print(os.path.basename(rel), sha(dst))

    rgb = np.asarray(Image.open(os.path.join(REF, "testdata/sample.jpg")).convert("RGB"), dtype=np.float64)
    r16, g16, b16 = rgb[..., 0] * 257.0, rgb[..., 1] * 257.0, rgb[..., 2] * 257.0
    gray = ((0.299 * r16 + 0.587 * g16 + 0.114 * b16) / 256.0).astype(np.uint8)  # truncation, like Go's uint8()
    assert gray.shape == (400, 320), gray.shape  # Rows=400, Cols=320
    out = os.path.join(DATA, "sample_gray_320x400.bin")
    gray.tofile(out)
    print("facefinder", sha(os.path.join(DATA, "facefinder")))
    print("sample_gray_320x400.bin", sha(out), gray.shape)


if __name__ == "__main__":
    main()
