"""Seeded synthetic grayscale frames for the parity tests and the benchmark.

Two value distributions, both fully determined by ``(seed, frame_index, rows, cols)``:

* ``SYN-NOISE``  iid uniform bytes from splitmix64(seed ^ frame_index) -- worst case for the
  cascade's early-out (mean 2.66 trees/window, 0 detections).
* ``SYN-FACES``  low-frequency background (8x8 block-upsampled SYN-NOISE) with K copies of the
  ``sample_gray`` fixture pasted at seeded, non-periodic positions at x1/2, x1 or x2 nearest-neighbour
  zoom -- natural-image reject profile (~2.0 trees/window) and O(10^2) raw detections per 1080p
  frame, so the detection-emit, order-restore and clustering kernels are exercised.

There is no network and no dataset: every "1080p frame" in tests/ and bench.py comes from here.
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_MASK = (1 << 64) - 1


def _splitmix64_block(seed: int, n_words: int) -> np.ndarray:
    """n_words successive outputs of splitmix64 seeded with `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n_words + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _frame_seed(seed: int, frame_index: int) -> int:
    # decorrelate consecutive frame indices before they reach splitmix64's additive counter
    return (seed ^ ((frame_index + 1) * 0xD1342543DE82EF95)) & _MASK


def syn_noise(rows: int, cols: int, seed: int = 1234, frame_index: int = 0) -> np.ndarray:
    """uint8 [rows, cols] of iid uniform bytes."""
    n = rows * cols
    words = _splitmix64_block(_frame_seed(seed, frame_index), (n + 7) // 8)
    return words.view(np.uint8)[:n].reshape(rows, cols).copy()


def sample_gray() -> np.ndarray:
    """The committed 400x320 (rows x cols) gray fixture made from the reference's testdata/sample.jpg
    (tests/golden/make_fixtures.py)."""
    return np.fromfile(os.path.join(_DATA, "sample_gray_320x400.bin"), dtype=np.uint8).reshape(400, 320)


def facefinder_bytes() -> bytes:
    """The reference's face cascade (model data), pigo_amd/data/facefinder."""
    with open(os.path.join(_DATA, "facefinder"), "rb") as f:
        return f.read()


def syn_faces(rows: int, cols: int, seed: int = 1234, frame_index: int = 0, n_faces: int = None, rotate_deg: float = 0.0) -> np.ndarray:
    """uint8 [rows, cols]: block-noise background + pasted face patches (see module docstring).  rotate_deg != 0: the patch is
    rotated first (scipy.ndimage.rotate, bilinear, reshape) -- -79 degrees is what a scan at angle 0.8 detects, so that the
    rotated path is benchmarked with as many survivors as the upright one (bench.py --face-rotation)."""
    fs = _frame_seed(seed, frame_index)
    br, bc = (rows + 7) // 8, (cols + 7) // 8
    coarse = _splitmix64_block(fs ^ 0xA5A5A5A5, (br * bc + 7) // 8).view(np.uint8)[: br * bc].reshape(br, bc)
    img = np.repeat(np.repeat(coarse, 8, axis=0), 8, axis=1)[:rows, :cols].copy()
    patch = sample_gray()
    if rotate_deg != 0.0:
        from scipy import ndimage
        patch = ndimage.rotate(patch, rotate_deg, reshape=True, order=1, mode="nearest")
    if n_faces is None:
        n_faces = max(1, int(round(18 * (rows * cols) / (1080.0 * 1920.0))))
    rnd = _splitmix64_block(fs ^ 0x5EED5EED, 3 * n_faces)
    zooms = (0.5, 1.0, 1.0, 2.0)
    for k in range(n_faces):
        z = zooms[int(rnd[3 * k] % np.uint64(4))]
        ph, pw = int(patch.shape[0] * z), int(patch.shape[1] * z)
        while ph > rows or pw > cols:  # shrink until it fits small frames
            z *= 0.5
            ph, pw = int(patch.shape[0] * z), int(patch.shape[1] * z)
        if ph < 8 or pw < 8:
            continue
        yy = (np.arange(ph) / z).astype(np.int64).clip(0, patch.shape[0] - 1)
        xx = (np.arange(pw) / z).astype(np.int64).clip(0, patch.shape[1] - 1)
        p = patch[yy][:, xx]
        r0 = int(rnd[3 * k + 1] % np.uint64(rows - ph + 1))
        c0 = int(rnd[3 * k + 2] % np.uint64(cols - pw + 1))
        img[r0:r0 + ph, c0:c0 + pw] = p
    return img


def make_frames(kind: str, n: int, rows: int, cols: int, seed: int = 1234, first_index: int = 0, rotate_deg: float = 0.0) -> np.ndarray:
    """uint8 [n, rows, cols] batch; frame f uses frame_index first_index + f."""
    out = np.empty((n, rows, cols), dtype=np.uint8)
    for f in range(n):
        out[f] = syn_noise(rows, cols, seed, first_index + f) if kind == "noise" else syn_faces(rows, cols, seed, first_index + f, rotate_deg=rotate_deg)
    return out


def syn_rgba(rows: int, cols: int, seed: int = 1234, frame_index: int = 0, opaque_rows: int = None) -> np.ndarray:
    """Seeded {R,G,B,A} test frame [rows, cols, 4] for RgbToGrayscale (core/grayscale.go:8-23): uniform random bytes;
    the first ``opaque_rows`` rows (default: all) get A=255 like a decoded JPEG, the rest keep their random alpha."""
    a = syn_noise(rows, cols * 4, seed=seed ^ 0x5A5A, frame_index=frame_index).reshape(rows, cols, 4).copy()
    k = rows if opaque_rows is None else opaque_rows
    a[:k, :, 3] = 255
    return a


def syn_uniform32(n: int, seed: int = 1234, index: int = 0) -> np.ndarray:
    """n seeded float32 values in [0, 1) (24 random bits each) -- stands in for math/rand's rand.Float32() stream that
    RunDetector draws its perturbations from (core/puploc.go:248-250)."""
    words = _splitmix64_block(_frame_seed(seed ^ 0xF10A7, index), n)
    return ((words >> np.uint64(40)).astype(np.float32) / np.float32(1 << 24)).astype(np.float32)


def cascade_bytes(name: str) -> bytes:
    """A cascade file shipped as data: "facefinder", "puploc", "lps/lp42", ..."""
    with open(os.path.join(_DATA, *name.split("/")), "rb") as fh:
        return fh.read()
