#!/usr/bin/env python3
"""Inputs for tests/golden/gen_golden.go (the real Go package): raw images + manifest.json under tests/golden/go_inputs/.

The scan cases are exactly those of golden_cases.json (make_golden.py); on top of them tie-heavy detection lists for
sort.Slice / ClusterDetections, RunDetector requests with a seeded math/rand and RgbToGrayscale vectors.  Nothing here needs
Go or a GPU; the directory is scratch (git-ignored) and is recreated by this script.

    python tests/golden/make_go_inputs.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pigo_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "go_inputs")


def f32hex(x):
    return np.float32(x).tobytes().hex()


def tie_lists():
    """Detection lists with many equal Q values: the order Go's (unstable) pdqsort leaves them in is observable."""
    rng = np.random.default_rng(20240921)
    out = []
    for name, n, levels, iou in (("ties_12", 12, 3, 0.2), ("ties_13", 13, 3, 0.2), ("ties_50", 50, 5, 0.1), ("ties_317", 317, 40, 0.2),
                                 ("ties_1500", 1500, 9, 0.3), ("ties_3000", 3000, 25, 0.0), ("distinct_200", 200, 10**7, 0.15)):
        rows = rng.integers(20, 1000, n)
        cols = rng.integers(20, 1800, n)
        scales = rng.choice([24, 40, 60, 90, 140], n)
        q = (rng.integers(1, levels + 1, n) / np.float32(3.0)).astype(np.float32)
        out.append({"name": name, "iou": iou, "dets": [[int(rows[i]), int(cols[i]), int(scales[i]), f32hex(q[i])] for i in range(n)]})
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(HERE, "golden_cases.json")) as fh:
        golden = json.load(fh)
    from conftest import golden_image
    images = {}
    scan = []
    for c in golden["cases"]:
        img = golden_image(c["name"])
        key = {"sample": "sample", "land": "land"}.get(c["name"].split("_")[0], c["name"] if not c["name"].startswith("faces_480x640") else "faces_480x640")
        if key not in images:
            images[key] = img
            img.tofile(os.path.join(OUT, key + ".bin"))
        assert images[key].shape == img.shape and (images[key] == img).all()
        scan.append({k: c[k] for k in ("name", "rows", "cols", "dim", "min_size", "max_size", "shift", "scale", "angle", "iou")} | {"file": key + ".bin"})
    # BASELINE config 2 / 4 on one seeded 1080p frame: ~4.1 M windows, ties among the detections
    f1080 = synth.make_frames("faces", 1, 1080, 1920, seed=1234)[0]
    f1080.tofile(os.path.join(OUT, "faces_1080p.bin"))
    for name, angle in (("faces_1080p", 0.0), ("faces_1080p_rot0.8", 0.8)):
        scan.append({"name": name, "file": "faces_1080p.bin", "rows": 1080, "cols": 1920, "dim": 1920, "min_size": 20, "max_size": 1000,
                     "shift": 0.1, "scale": 1.1, "angle": angle, "iou": 0.2})
    pup = []
    for k, (casc, row, col, scale, angle, flip) in enumerate((("puploc", 187, 109, 65.25, 0.0, False), ("puploc", 187, 199, 65.25, 0.0, True),
                                                              ("puploc", 190, 110, 40.0, 0.4, False), ("lps/lp42", 230, 150, 120.5, 0.0, False),
                                                              ("lps/lp93", 260, 160, 110.0, 0.0, True))):
        pup.append({"name": f"pup_{k}_{casc.replace('/', '_')}", "cascade": casc, "file": "sample.bin", "rows": 400, "cols": 320, "dim": 320,
                    "row": row, "col": col, "scale": f32hex(scale), "perturbs": 63, "angle": angle, "flip_v": flip, "seed": 42 + k})
    gray = []
    rgba = synth.syn_rgba(37, 53, seed=11, opaque_rows=20)
    rgba.tofile(os.path.join(OUT, "rgba_37x53.bin"))
    for kind in ("NRGBA", "RGBA"):
        gray.append({"name": f"rgba_37x53_{kind}", "file": "rgba_37x53.bin", "width": 53, "height": 37, "kind": kind})
    with open(os.path.join(OUT, "manifest.json"), "w") as fh:
        json.dump({"scan": scan, "lists": tie_lists(), "puploc": pup, "gray": gray}, fh)
    print("wrote", OUT, f"({len(scan)} scan cases, {len(pup)} RunDetector cases)")


if __name__ == "__main__":
    main()
