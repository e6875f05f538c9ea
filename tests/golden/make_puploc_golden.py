#!/usr/bin/env python3
"""Mint tests/golden/puploc_golden.json: known answers for RunDetector / GetLandmarkPoint (core/puploc.go:239-277,
core/flploc.go:36-57) from the C oracle, written only if the independent NumPy restatement agrees on every value.
PARITY UNPINNED against the Go binary (no Go toolchain here).  Perturbation randoms are seeded
(pigo_amd.synth.syn_uniform32) -- the reference draws them from the global math/rand source.

    python tests/golden/make_puploc_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import oracle  # noqa: E402
from oracle.np_restatement import NpPuploc  # noqa: E402
from pigo_amd import synth  # noqa: E402
import landmark_pipeline as lp  # noqa: E402


def f32hex(x):
    return np.float32(x).tobytes().hex()


# cascade, image, row, col, scale, perturbs, angle, flip_v, pool ("fresh" | "chain": reuse the previous case's pool)
CASES = [
    ("puploc", "sample", 187, 109, 65.25, 63, 0.0, False, "fresh"),
    ("puploc", "sample", 187, 202, 65.25, 50, 0.0, False, "fresh"),   # Perturbs < 63 on a new pool: 13 zeros are sorted in
    ("puploc", "sample", 187, 202, 65.25, 50, 0.0, False, "chain"),   # ... and on a used one: 13 stale values instead
    ("puploc", "sample", 187, 109, 65.25, 63, 0.3, False, "fresh"),   # classifyRotatedRegion
    ("puploc", "sample", 187, 109, 65.25, 63, 1.7, True, "fresh"),    # angle clamped to 1.0, flipV
    ("puploc", "sample", 3, 2, 40.0, 63, 0.0, True, "fresh"),         # clamps at the image border
    ("puploc", "sample", 398, 318, 90.0, 63, 0.04, False, "fresh"),   # k=1, far corner
    ("puploc", "noise", 100, 100, 20.0, 63, 0.0, False, "fresh"),
    ("puploc", "noise", 10, 10, 20.0, 50, 0.0, False, "fresh"),       # BenchmarkPuplocDetectorRun's request (puploc_test.go:104)
    ("lps/lp42", "sample", 200, 150, 120.0, 63, 0.0, False, "fresh"),
    ("lps/lp42", "sample", 200, 150, 120.0, 63, 0.0, True, "fresh"),
    ("lps/lp84", "sample", 230, 160, 110.0, 63, 0.0, True, "chain"),
    ("lps/lp93", "noise", 120, 160, 75.5, 1, 0.0, False, "fresh"),
    ("lps/lp93", "noise", 120, 160, 75.5, 0, 0.0, False, "fresh"),    # no perturbation at all: median of the pool
]


def image(name):
    if name == "sample":
        return synth.sample_gray()
    return synth.syn_noise(240, 320, seed=1234, frame_index=3)


def main():
    out = {"_doc": "RunDetector / GetLandmarkPoint known answers; minted by tests/golden/make_puploc_golden.py from oracle/pigo_oracle.c "
                   "after agreement with oracle/np_restatement.py; PARITY UNPINNED against the Go binary", "cases": []}
    casc_o, casc_n = {}, {}
    pool_o = pool_n = None
    for i, (cname, iname, row, col, scale, P, angle, flip, pmode) in enumerate(CASES):
        if cname not in casc_o:
            pk = synth.cascade_bytes(cname)
            casc_o[cname], casc_n[cname] = oracle.OraclePuploc.unpack(pk), NpPuploc(pk)
        img = image(iname)
        rows, cols = img.shape
        rnd = synth.syn_uniform32(3 * 63, seed=1234, index=100 + i)
        if pmode == "fresh" or pool_o is None:
            pool_o, pool_n = np.zeros((3, 63), np.float32), np.zeros((3, 63), np.float32)
        a = casc_o[cname].run_detector(row, col, scale, P, img, rows, cols, cols, angle, flip, rnd, pool_o)
        b = casc_n[cname].run_detector(row, col, scale, P, img, rows, cols, cols, angle, flip, rnd, pool_n)
        assert a == b and (pool_o == pool_n).all(), (i, a, b)
        out["cases"].append({"cascade": cname, "image": iname, "row": row, "col": col, "scale": scale, "perturbs": P, "angle": angle,
                             "flip_v": flip, "pool": pmode, "rnd_index": 100 + i,
                             "want": [a[0], a[1], f32hex(a[2])], "pool_after_sha": __import__("hashlib").sha256(pool_o.tobytes()).hexdigest()})
    # the reference's landmark test sequence on the fixture's face (core/flploc_test.go:84-153)
    gray = synth.sample_gray()
    o = oracle.OraclePigo.unpack(synth.facefinder_bytes())
    dets = o.run_cascade(gray, 400, 320, 320, 20, 1000, 0.1, 1.1, 0.0)       # cParams of core/pigo_test.go:44-50
    faces = [d for d in o.cluster_detections(dets, 0.1) if d["scale"] > 50]  # iou 0.1, Scale > 50 (flploc_test.go:100,104)
    assert len(faces) == 1
    face = (int(faces[0]["row"]), int(faces[0]["col"]), int(faces[0]["scale"]))
    so = lp.run_sequence(lp.OracleBackend(gray, 400, 320, 320), face)
    sn = lp.run_sequence(lp.NumpyBackend(gray, 400, 320, 320), face)
    assert so["left"] == sn["left"] and so["right"] == sn["right"]
    for x, y in zip(so["points"], sn["points"]):
        assert x == y, (x, y)
    assert sum(1 for _, _, p in so["points"] if p[0] > 0 and p[1] > 0) == 15  # the reference's own invariant (flploc_test.go:150-153)
    out["sequence"] = {"face": list(face), "seed": 1234,
                       "left": [so["left"][0], so["left"][1], f32hex(so["left"][2])],
                       "right": [so["right"][0], so["right"][1], f32hex(so["right"][2])],
                       "points": [[n, fl, p[0], p[1], f32hex(p[2])] for n, fl, p in so["points"]]}
    with open(os.path.join(HERE, "puploc_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", len(out["cases"]), "cases; face", face, "eyes", so["left"], so["right"])
    for n, fl, p in so["points"]:
        print(" ", n, fl, p)


if __name__ == "__main__":
    main()
