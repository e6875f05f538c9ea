#!/usr/bin/env python3
"""Mint the golden vectors in tests/golden/*.json.

The reference's own tests pin almost nothing numeric on this path and the Go code
cannot run here, so the goldens are minted from the C oracle (oracle/pigo_oracle.c) and written ONLY
if the independent NumPy restatement (oracle/np_restatement.py) reproduces every value bit-for-bit.
PARITY UNPINNED against the Go binary itself; see DESIGN.md.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import oracle  # noqa: E402
from oracle.np_restatement import NpPigo  # noqa: E402
from pigo_amd import synth  # noqa: E402


def f32hex(x):
    return np.float32(x).tobytes().hex()


def case(o, n, name, img, rows, cols, dim, mn, mx, shift, scale, angle, iou):
    d, nwin = None, None
    d, nwin, _ = o.run_cascade(img, rows, cols, dim, mn, mx, shift, scale, angle, want_stats=True)
    d2, nwin2 = n.run_cascade(img, rows, cols, dim, mn, mx, shift, scale, angle)
    assert nwin == nwin2 and len(d) == len(d2), (name, nwin, nwin2, len(d), len(d2))
    for a, b in zip(d, d2):
        assert (int(a["row"]), int(a["col"]), int(a["scale"])) == tuple(int(v) for v in b[:3]), name
        assert np.float32(b[3]) == a["q"], name
    raw = [[int(a["row"]), int(a["col"]), int(a["scale"]), f32hex(a["q"])] for a in d]
    dd = d.copy()
    cl, ties = o.cluster_detections(dd, iou, want_ties=True)
    if ties == 0 or len(d) <= 12:  # the stable NumPy sort equals Go's pdqsort only then
        _, cl2 = NpPigo.cluster_detections([(a["row"], a["col"], a["scale"], a["q"]) for a in d], iou)
        assert len(cl) == len(cl2), name
        for a, b in zip(cl, cl2):
            assert (int(a["row"]), int(a["col"]), int(a["scale"])) == tuple(int(v) for v in b[:3]) and np.float32(b[3]) == a["q"], name
    clusters = [[int(a["row"]), int(a["col"]), int(a["scale"]), f32hex(a["q"])] for a in cl]
    print(f"{name}: {nwin} windows, {len(raw)} detections, {len(clusters)} clusters, ties={ties}")
    return {"name": name, "rows": rows, "cols": cols, "dim": dim, "min_size": mn, "max_size": mx, "shift": shift, "scale": scale,
            "angle": angle, "iou": iou, "windows": nwin, "detections": raw, "clusters": clusters, "ties": ties}


def main():
    packet = synth.facefinder_bytes()
    o = oracle.OraclePigo.unpack(packet)
    n = NpPigo.unpack(packet)
    gray = synth.sample_gray()
    cases = []
    # the reference's own parameter sets on the sample fixture
    cases.append(case(o, n, "sample_tests_0.2", gray, 400, 320, 320, 20, 1000, 0.2, 1.1, 0.0, 0.1))    # core/pigo_test.go:44-50
    cases.append(case(o, n, "sample_readme_0.1", gray, 400, 320, 320, 20, 1000, 0.1, 1.1, 0.0, 0.2))   # README.md:97-126
    cases.append(case(o, n, "sample_cli_defaults", gray, 400, 320, 320, 20, 1000, 0.15, 1.15, 0.0, 0.15))  # cmd/pigo/main.go:108-113
    cases.append(case(o, n, "sample_iou0", gray, 400, 320, 320, 20, 1000, 0.1, 1.1, 0.0, 0.0))         # examples/*: iou 0
    # rotated path on a landscape crop of the fixture (rows < cols so quirk Q1 cannot leave the slice)
    land = np.ascontiguousarray(gray[40:340, :])  # 300 x 320
    cases.append(case(o, n, "land_rot_0.03", land, 300, 320, 320, 20, 1000, 0.1, 1.1, 0.03, 0.1))
    cases.append(case(o, n, "land_rot_0.8", land, 300, 320, 320, 20, 1000, 0.1, 1.1, 0.8, 0.01))       # README.md:39
    cases.append(case(o, n, "land_rot_1.0", land, 300, 320, 320, 20, 1000, 0.1, 1.1, 1.0, 0.1))
    cases.append(case(o, n, "land_rot_1.7", land, 300, 320, 320, 20, 1000, 0.1, 1.1, 1.7, 0.1))        # clamped to 1.0, pigo.go:233
    # seeded synthetic frames (small, so the NumPy cross-check stays fast)
    f = synth.syn_faces(480, 640, seed=1234, frame_index=0)
    cases.append(case(o, n, "faces_480x640", f, 480, 640, 640, 20, 1000, 0.1, 1.1, 0.0, 0.2))
    cases.append(case(o, n, "faces_480x640_wasm", f, 480, 640, 640, 200, 480, 0.1, 1.1, 0.0, 0.1))     # wasm/detector/detector.go:156-169
    cases.append(case(o, n, "faces_480x640_facedet", f, 480, 640, 640, 100, 600, 0.15, 1.1, 0.0, 0.0))  # examples/facedet/pigo.go:62-66
    cases.append(case(o, n, "faces_480x640_rot0.8", f, 480, 640, 640, 20, 1000, 0.1, 1.1, 0.8, 0.01))
    nz = synth.syn_noise(240, 320, seed=1234, frame_index=3)
    cases.append(case(o, n, "noise_240x320", nz, 240, 320, 320, 20, 1000, 0.1, 1.1, 0.0, 0.2))
    strided = np.zeros((200, 300), dtype=np.uint8)
    strided[:, :260] = synth.syn_faces(200, 260, seed=7, frame_index=1)
    strided[:, 260:] = 255  # padding columns must never influence the result
    cases.append(case(o, n, "faces_200x260_dim300", strided, 200, 260, 300, 20, 1000, 0.1, 1.1, 0.0, 0.2))
    with open(os.path.join(HERE, "golden_cases.json"), "w") as fh:
        json.dump({"facefinder_sha256": "d8014993e7298c7b1865d1f8b855d6dbf4ec5c808bf879e2091ab6837abf90cd", "cases": cases}, fh, indent=1)


if __name__ == "__main__":
    main()
