"""ctypes loader for the CPU oracle (oracle/pigo_oracle.c).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
import this package; nothing under pigo_amd/ does.  PARITY UNPINNED (see pigo_oracle.c header): the
reference is pure Go and cannot be executed in this image.

The Python surface mirrors the reference API (core/pigo.go): ``OraclePigo.unpack`` /
``run_cascade`` / ``cluster_detections`` with detections as numpy structured arrays
``(row i8, col i8, scale i8, q f4)``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpigo_oracle.so")

#: Detection, core/pigo.go:195-200 (Go/amd64 layout: 3 x int64 + float32 + 4 pad bytes = 32 B)
DET_DTYPE = np.dtype({"names": ["row", "col", "scale", "q"], "formats": ["<i8", "<i8", "<i8", "<f4"],
                      "offsets": [0, 8, 16, 24], "itemsize": 32})

ERR_PANIC = -1


class OraclePanic(RuntimeError):
    """The reference Go code would panic (slice index out of range) on this input."""


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pigo_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpigo_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        ll, vp, dbl = C.c_longlong, C.c_void_p, C.c_double
        L.oracle_unpack.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
        L.oracle_unpack.restype = C.c_int
        L.oracle_free.argtypes = [vp]
        L.oracle_free.restype = None
        L.oracle_tree_depth.argtypes = [vp]
        L.oracle_tree_depth.restype = C.c_uint32
        L.oracle_tree_num.argtypes = [vp]
        L.oracle_tree_num.restype = C.c_uint32
        for name, typ in (("oracle_tree_codes", C.POINTER(C.c_int8)), ("oracle_tree_pred", C.POINTER(C.c_float)),
                          ("oracle_tree_threshold", C.POINTER(C.c_float))):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = typ
        L.oracle_classify_region.argtypes = [vp, ll, ll, ll, vp, ll, ll, C.POINTER(C.c_int)]
        L.oracle_classify_region.restype = C.c_float
        L.oracle_classify_rotated_region.argtypes = [vp, ll, ll, ll, dbl, ll, ll, vp, ll, ll, C.POINTER(C.c_int)]
        L.oracle_classify_rotated_region.restype = C.c_float
        L.oracle_run_cascade.argtypes = [vp, vp, ll, ll, ll, ll, ll, ll, dbl, dbl, dbl, vp, ll, C.POINTER(ll), vp]
        L.oracle_run_cascade.restype = ll
        L.oracle_sort_by_q.argtypes = [vp, ll]
        L.oracle_sort_by_q.restype = None
        L.oracle_calc_iou.argtypes = [vp, vp]
        L.oracle_calc_iou.restype = dbl
        L.oracle_cluster_detections.argtypes = [vp, ll, dbl, vp, C.POINTER(ll)]
        L.oracle_cluster_detections.restype = ll
        L.oracle_rgb_to_grayscale.argtypes = [vp, ll, ll, ll, C.c_int, vp]
        L.oracle_rgb_to_grayscale.restype = C.c_int
        L.oracle_puploc_unpack.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
        L.oracle_puploc_unpack.restype = C.c_int
        L.oracle_puploc_free.argtypes = [vp]
        L.oracle_puploc_free.restype = None
        for name, typ in (("oracle_puploc_stages", C.c_uint32), ("oracle_puploc_trees", C.c_uint32), ("oracle_puploc_depth", C.c_uint32),
                          ("oracle_puploc_scales", C.c_float), ("oracle_puploc_codes", C.POINTER(C.c_int8)),
                          ("oracle_puploc_preds", C.POINTER(C.c_float))):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = typ
        L.oracle_puploc_classify.argtypes = [vp, C.c_float, C.c_float, C.c_float, dbl, C.c_int, ll, ll, vp, ll, ll, C.c_int, vp]
        L.oracle_puploc_classify.restype = C.c_int
        L.oracle_puploc_run_detector.argtypes = [vp, vp, vp, ll, ll, ll, ll, dbl, C.c_int, vp, vp, vp]
        L.oracle_puploc_run_detector.restype = C.c_int
        L.oracle_get_landmark_point.argtypes = [vp, vp, vp, vp, ll, ll, ll, ll, ll, C.c_int, vp, vp, vp]
        L.oracle_get_landmark_point.restype = C.c_int
        _lib = L
    return _lib


class OraclePigo:
    """Restatement of ``type Pigo`` (core/pigo.go:37-43) backed by the C oracle."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_free(self._h)
            self._h = None

    # -- Unpack, core/pigo.go:51 ------------------------------------------------------------------
    @classmethod
    def unpack(cls, packet: bytes):
        h = C.c_void_p()
        rc = lib().oracle_unpack(packet, len(packet), C.byref(h))
        if rc == ERR_PANIC:
            raise OraclePanic("Unpack: packet too short (Go would panic)")
        if rc != 0:
            raise MemoryError("oracle_unpack failed")
        return cls(h)

    @property
    def tree_depth(self):
        return int(lib().oracle_tree_depth(self._h))

    @property
    def tree_num(self):
        return int(lib().oracle_tree_num(self._h))

    def tables(self):
        """(treeCodes int8[ntrees, 4*2^d], treePred f32[ntrees, 2^d], treeThreshold f32[ntrees])"""
        n, d = self.tree_num, self.tree_depth
        L = lib()
        codes = np.ctypeslib.as_array(L.oracle_tree_codes(self._h), shape=(n * 4 * (1 << d),)).copy()
        pred = np.ctypeslib.as_array(L.oracle_tree_pred(self._h), shape=(n * (1 << d),)).copy()
        thr = np.ctypeslib.as_array(L.oracle_tree_threshold(self._h), shape=(n,)).copy()
        return codes.reshape(n, 4 << d), pred.reshape(n, 1 << d), thr

    # -- classifyRegion / classifyRotatedRegion ----------------------------------------------------
    def classify_region(self, r, c, s, pixels, dim):
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8).ravel()
        panic = C.c_int(0)
        q = lib().oracle_classify_region(self._h, r, c, s, pixels.ctypes.data, pixels.size, dim, C.byref(panic))
        if panic.value:
            raise OraclePanic("classifyRegion: index out of range")
        return np.float32(q)

    def classify_rotated_region(self, r, c, s, a, nrows, ncols, pixels, dim):
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8).ravel()
        panic = C.c_int(0)
        q = lib().oracle_classify_rotated_region(self._h, r, c, s, a, nrows, ncols, pixels.ctypes.data, pixels.size, dim,
                                                 C.byref(panic))
        if panic.value:
            raise OraclePanic("classifyRotatedRegion: index out of range")
        return np.float32(q)

    # -- RunCascade, core/pigo.go:212 -------------------------------------------------------------
    def run_cascade(self, pixels, rows, cols, dim, min_size, max_size, shift_factor, scale_factor, angle=0.0,
                    want_stats=False):
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8).ravel()
        cap = 1 << 12
        nwin = C.c_longlong(0)
        hist = np.zeros(self.tree_num + 1, dtype=np.uint64) if want_stats else None
        while True:
            out = np.zeros(cap, dtype=DET_DTYPE)
            if hist is not None:
                hist[:] = 0
            n = lib().oracle_run_cascade(self._h, pixels.ctypes.data, pixels.size, rows, cols, dim, min_size, max_size,
                                         shift_factor, scale_factor, angle, out.ctypes.data, cap, C.byref(nwin),
                                         hist.ctypes.data if hist is not None else None)
            if n == ERR_PANIC:
                raise OraclePanic("RunCascade: index out of range")
            if n <= cap:
                break
            cap = int(n)
        dets = out[: int(n)].copy()
        if want_stats:
            return dets, int(nwin.value), hist
        return dets

    # -- ClusterDetections, core/pigo.go:262 ------------------------------------------------------
    def cluster_detections(self, dets, iou_threshold, want_ties=False):
        """Sorts ``dets`` IN PLACE (like the reference) and returns the clusters."""
        assert dets.dtype == DET_DTYPE and dets.flags.c_contiguous
        n = len(dets)
        out = np.zeros(max(n, 1), dtype=DET_DTYPE)
        ties = C.c_longlong(0)
        k = lib().oracle_cluster_detections(dets.ctypes.data, n, iou_threshold, out.ctypes.data, C.byref(ties))
        res = out[: int(k)].copy()
        if want_ties:
            return res, int(ties.value)
        return res


def sort_by_q(dets):
    """Go's sort.Slice by ascending Q (pdqsort restatement), in place."""
    assert dets.dtype == DET_DTYPE and dets.flags.c_contiguous
    lib().oracle_sort_by_q(dets.ctypes.data, len(dets))
    return dets


def make_dets(rows):
    """list of (row, col, scale, q) -> structured array"""
    a = np.zeros(len(rows), dtype=DET_DTYPE)
    for i, (r, c, s, q) in enumerate(rows):
        a[i] = (r, c, s, q)
    return a


#: pixel kinds of rgb_to_grayscale: *image.NRGBA, *image.RGBA, the wasm canvas variant
PIX_NRGBA, PIX_RGBA, PIX_CANVAS = 0, 1, 2


def rgb_to_grayscale(pix, kind=PIX_NRGBA):
    """RgbToGrayscale (core/grayscale.go:8-23) on an (H, W, 4) uint8 array whose rows may be strided."""
    pix = np.asarray(pix, dtype=np.uint8)
    assert pix.ndim == 3 and pix.shape[2] == 4 and pix.strides[2] == 1 and pix.strides[1] == 4
    h, w = pix.shape[:2]
    out = np.zeros(h * w, dtype=np.uint8)
    if h and w:
        rc = lib().oracle_rgb_to_grayscale(pix.ctypes.data, w, h, pix.strides[0], kind, out.ctypes.data)
        if rc != 0:
            raise ValueError("unknown pixel kind %r" % (kind,))
    return out


class _PuplocDet(C.Structure):
    """Puploc, core/puploc.go:14-19 (Go/amd64 layout: int, int, float32, int)"""
    _fields_ = [("row", C.c_longlong), ("col", C.c_longlong), ("scale", C.c_float), ("perturbs", C.c_longlong)]


class OraclePuploc:
    """Restatement of ``type PuplocCascade`` (core/puploc.go:22-29) backed by the C oracle.

    ``rnd`` everywhere = the 3*Perturbs values ``rand.Float32()`` would hand RunDetector (puploc.go:248-250), float32
    in draw order; ``pool`` = the sync.Pool object's three 63-entry arrays (float32 [3, 63], modified in place) or
    None for a brand-new one."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_puploc_free(self._h)
            self._h = None

    @classmethod
    def unpack(cls, packet: bytes):  # UnpackCascade, puploc.go:38
        h = C.c_void_p()
        rc = lib().oracle_puploc_unpack(packet, len(packet), C.byref(h))
        if rc == ERR_PANIC:
            raise OraclePanic("UnpackCascade: packet too short (Go would panic)")
        if rc != 0:
            raise MemoryError("oracle_puploc_unpack failed")
        return cls(h)

    @property
    def header(self):
        L = lib()
        return (int(L.oracle_puploc_stages(self._h)), float(L.oracle_puploc_scales(self._h)), int(L.oracle_puploc_trees(self._h)),
                int(L.oracle_puploc_depth(self._h)))

    def tables(self):
        st, _, tr, d = self.header
        L = lib()
        nc, npred = st * tr * (4 * (1 << d) - 4), st * tr * (1 << d) * 2
        return (np.ctypeslib.as_array(L.oracle_puploc_codes(self._h), shape=(nc,)).copy(),
                np.ctypeslib.as_array(L.oracle_puploc_preds(self._h), shape=(npred,)).copy())

    def classify(self, r, c, s, pixels, rows, cols, dim, angle=0.0, rotated=False, flip_v=False):
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8).ravel()
        res = np.zeros(3, dtype=np.float32)
        rc = lib().oracle_puploc_classify(self._h, r, c, s, angle, int(rotated), rows, cols, pixels.ctypes.data, pixels.size, dim,
                                          int(flip_v), res.ctypes.data)
        if rc == ERR_PANIC:
            raise OraclePanic("puploc classifyRegion: index out of range")
        return res

    @staticmethod
    def _pool(pool):
        if pool is None:
            return None
        assert pool.dtype == np.float32 and pool.shape == (3, 63) and pool.flags.c_contiguous
        return pool.ctypes.data

    def run_detector(self, row, col, scale, perturbs, pixels, rows, cols, dim, angle, flip_v, rnd, pool=None):
        """RunDetector, puploc.go:239 -> (row, col, scale)"""
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8).ravel()
        rnd = np.ascontiguousarray(rnd, dtype=np.float32).ravel()
        assert rnd.size >= 3 * max(min(perturbs, 63), 0)
        pl, out = _PuplocDet(row, col, scale, perturbs), _PuplocDet()
        rc = lib().oracle_puploc_run_detector(self._h, C.byref(pl), pixels.ctypes.data, pixels.size, rows, cols, dim, angle, int(flip_v),
                                              rnd.ctypes.data, self._pool(pool), C.byref(out))
        if rc == ERR_PANIC:
            raise OraclePanic("RunDetector: index out of range")
        return int(out.row), int(out.col), np.float32(out.scale)

    def get_landmark_point(self, left_eye, right_eye, pixels, rows, cols, dim, perturb, flip_v, rnd, pool=None):
        """GetLandmarkPoint, flploc.go:36; eyes are (row, col, scale) triples -> (row, col, scale)"""
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8).ravel()
        rnd = np.ascontiguousarray(rnd, dtype=np.float32).ravel()
        le, re, out = _PuplocDet(left_eye[0], left_eye[1], left_eye[2], 0), _PuplocDet(right_eye[0], right_eye[1], right_eye[2], 0), _PuplocDet()
        rc = lib().oracle_get_landmark_point(self._h, C.byref(le), C.byref(re), pixels.ctypes.data, pixels.size, rows, cols, dim, perturb,
                                             int(flip_v), rnd.ctypes.data, self._pool(pool), C.byref(out))
        if rc == ERR_PANIC:
            raise OraclePanic("GetLandmarkPoint: index out of range")
        return int(out.row), int(out.col), np.float32(out.scale)
