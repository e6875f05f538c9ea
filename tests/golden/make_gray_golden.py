#!/usr/bin/env python3
"""Mint tests/golden/gray_golden.json: known answers for RgbToGrayscale (core/grayscale.go:8-23).

Like make_golden.py: the Go code cannot run here (PARITY UNPINNED), so the vectors come from the C oracle and are
written only if the independent NumPy restatement agrees on every byte.  Inputs are seeded (pigo_amd.synth.syn_rgba),
outputs are stored as sha256 plus the first 16 bytes.

    python tests/golden/make_gray_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import oracle  # noqa: E402
from oracle.np_restatement import np_rgb_to_grayscale  # noqa: E402
from pigo_amd import synth  # noqa: E402

CASES = [  # name, rows, cols, frame_index, opaque_rows
    ("rgba_97x131_mixed_alpha", 97, 131, 0, 40),
    ("rgba_64x64_opaque", 64, 64, 1, None),
    ("rgba_240x320_mixed_alpha", 240, 320, 2, 120),
    ("rgba_1080x1920_opaque", 1080, 1920, 3, None),
]


def main():
    out = {"_doc": "RgbToGrayscale known answers, minted by tests/golden/make_gray_golden.py from oracle/pigo_oracle.c after "
                   "agreement with oracle/np_restatement.py; PARITY UNPINNED against the Go binary", "cases": []}
    for name, rows, cols, fi, op in CASES:
        img = synth.syn_rgba(rows, cols, seed=1234, frame_index=fi, opaque_rows=op)
        rec = {"name": name, "rows": rows, "cols": cols, "frame_index": fi, "opaque_rows": op, "kinds": {}}
        for kind in (0, 1, 2):
            a = oracle.rgb_to_grayscale(img, kind)
            b = np_rgb_to_grayscale(img, kind)
            assert a.shape == b.shape and (a == b).all(), (name, kind)
            rec["kinds"][str(kind)] = {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "head": [int(v) for v in a[:16]]}
        out["cases"].append(rec)
    with open(os.path.join(HERE, "gray_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
