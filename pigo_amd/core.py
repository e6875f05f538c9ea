"""Host-side mirror of the reference API (package github.com/esimov/pigo/core) over libpigo_hip.so.

The Go toolchain is not available in this image, so the drop-in shim a Go maintainer would add is shown
as source in INTEGRATION.md; this module is the same thin layer in Python (ctypes), with the reference's
names, argument meaning and error behaviour, so that the parity tests read like core/pigo_test.go:

    pg = Pigo().Unpack(cascade_bytes)                      # core/pigo.go:51
    cp = CascadeParams(MinSize=20, MaxSize=1000, ShiftFactor=0.1, ScaleFactor=1.1,
                       ImageParams=ImageParams(Pixels=gray, Rows=rows, Cols=cols, Dim=cols))
    dets = pg.RunCascade(cp, 0.0)                          # core/pigo.go:212
    dets = pg.ClusterDetections(dets, 0.2)                 # core/pigo.go:262

There is NO CPU implementation behind these calls: every method goes through the C ABI into the HIP
kernels and raises if the library or a GPU is missing.
"""
import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

from . import build as _build

#: Detection, core/pigo.go:195-200 -- the 16-byte wire record of the C ABI (pigo_det)
DET_DTYPE = np.dtype([("row", "<i4"), ("col", "<i4"), ("scale", "<i4"), ("q", "<f4")])

#: how RgbToGrayscale reads the 4-byte pixels: *image.NRGBA, *image.RGBA, the wasm canvas formula (include/pigo_hip.h)
PIX_NRGBA, PIX_RGBA, PIX_CANVAS = 0, 1, 2

PIGO_OK, ERR_PACKET, ERR_PARAM, ERR_HIP, ERR_CAPACITY, ERR_PANIC, ERR_NOMEM, ERR_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7


class PigoPanic(RuntimeError):
    """The reference Go code would panic here (slice index out of range)."""


class PigoError(RuntimeError):
    pass


class PlanInfo(C.Structure):
    _fields_ = [("windows_per_frame", C.c_int64), ("n_scales", C.c_int32), ("n_ladder", C.c_int32), ("tiles_per_frame", C.c_int32),
                ("n_head_trees", C.c_int32), ("variant", C.c_int32), ("max_frames", C.c_int32), ("det_cap", C.c_int32),
                ("queue_capacity", C.c_int64), ("workspace_bytes", C.c_int64)]


#: every symbol include/pigo_hip.h declares (tests check that the built library exports all of them)
ABI_SYMBOLS = [
    "pigo_last_error", "pigo_device_count", "pigo_cascade_create", "pigo_cascade_info", "pigo_cascade_tables", "pigo_cascade_destroy",
    "pigo_run_cascade", "pigo_cluster_detections", "pigo_sort_by_q", "pigo_plan_create", "pigo_plan_destroy", "pigo_plan_info",
    "pigo_plan_set_variant", "pigo_plan_run", "pigo_plan_cluster", "pigo_plan_status", "pigo_plan_last_flags", "pigo_plan_run_sync", "pigo_plan_set_profiling",
    "pigo_plan_last_timings", "pigo_plan_last_queue_count", "pigo_plan_debug_stats", "pigo_plan_debug_trace",
    "pigo_rgb_to_grayscale", "pigo_gray_batch",
    "pigo_puploc_create", "pigo_puploc_info", "pigo_puploc_destroy", "pigo_puploc_run_detector", "pigo_get_landmark_point",
    "pigo_puploc_run_batch", "pigo_puploc_status",
    "pigo_comm_unique_id", "pigo_comm_init", "pigo_comm_info", "pigo_comm_uses_rccl", "pigo_comm_abort", "pigo_comm_destroy", "pigo_shard_bounds", "pigo_wire_words", "pigo_wire_row_flags",
    "pigo_run_batch_sharded", "pigo_pack_lists", "pigo_unpack_list",
]

_lib = None


def library_path():
    """In-tree libpigo_hip.so; PIGO_HIP_LIB overrides it (A/B runs of experimental builds)."""
    return os.environ.get("PIGO_HIP_LIB") or _build.LIB


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch's ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; if
    libpigo_hip.so pulled in /opt/rocm's copy first, a later `import torch` would bring a second HSA runtime into the
    process and find no GPUs (and streams could not be shared).  So when torch is installed but not imported yet, its
    bundled runtime is loaded first (by path, without importing torch); libpigo_hip.so then binds to it by soname --
    the same situation as `import torch` before `import pigo_amd`."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system ROCm runtime


def load_library():
    """dlopen libpigo_hip.so (in-tree).  Fails loudly when it has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise PigoError(f"{path} is missing: build it with `python -m pigo_amd.build` (hipcc, gfx950). "
                        "pigo_amd has no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(path)
    vp, i32, dbl, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    L.pigo_last_error.restype = C.c_char_p
    L.pigo_device_count.restype = i32
    L.pigo_cascade_create.argtypes = [C.c_char_p, sz, i32, C.POINTER(vp)]
    L.pigo_cascade_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.pigo_cascade_tables.argtypes = [vp, vp, sz, vp, sz, vp, sz]
    L.pigo_cascade_destroy.argtypes = [vp]
    L.pigo_cascade_destroy.restype = None
    L.pigo_run_cascade.argtypes = [vp, vp, sz, i32, i32, i32, i32, i32, dbl, dbl, dbl, vp, i32, C.POINTER(i32)]
    L.pigo_cluster_detections.argtypes = [vp, vp, i32, dbl, vp, i32, C.POINTER(i32)]
    L.pigo_sort_by_q.argtypes = [vp, i32]
    L.pigo_sort_by_q.restype = None
    L.pigo_plan_create.argtypes = [vp, i32, i32, i32, i32, i32, dbl, dbl, dbl, i32, i32, C.POINTER(vp)]
    L.pigo_plan_destroy.argtypes = [vp]
    L.pigo_plan_destroy.restype = None
    L.pigo_plan_info.argtypes = [vp, C.POINTER(PlanInfo)]
    L.pigo_plan_set_variant.argtypes = [vp, i32]
    L.pigo_plan_run.argtypes = [vp, vp, sz, i32, vp, vp, vp]
    L.pigo_plan_run_sync.argtypes = [vp, vp, sz, i32, vp, vp, vp]
    L.pigo_plan_cluster.argtypes = [vp, vp, vp, i32, dbl, vp, vp, vp, vp, vp]
    L.pigo_plan_status.argtypes = [vp]
    L.pigo_plan_set_profiling.argtypes = [vp, i32]
    L.pigo_plan_last_timings.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), i32]
    L.pigo_plan_last_queue_count.argtypes = [vp, C.POINTER(C.c_int64)]
    L.pigo_plan_debug_stats.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    L.pigo_plan_debug_trace.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    L.pigo_rgb_to_grayscale.argtypes = [i32, vp, sz, i32, i32, i32, i32, vp, sz]
    L.pigo_gray_batch.argtypes = [i32, vp, sz, i32, i32, i32, i32, i32, vp, sz, i32, vp]
    L.pigo_puploc_create.argtypes = [C.c_char_p, sz, i32, C.POINTER(vp)]
    L.pigo_puploc_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.pigo_puploc_destroy.argtypes = [vp]
    L.pigo_puploc_destroy.restype = None
    L.pigo_puploc_run_detector.argtypes = [vp, vp, vp, sz, i32, i32, i32, dbl, i32, vp, vp, vp]
    L.pigo_get_landmark_point.argtypes = [vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, vp, vp, vp]
    L.pigo_puploc_run_batch.argtypes = [vp, vp, sz, i32, i32, i32, i32, dbl, vp, vp, vp, i32, vp, vp]
    L.pigo_puploc_status.argtypes = [vp]
    L.pigo_plan_last_flags.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.pigo_comm_unique_id.argtypes = [vp]
    L.pigo_comm_init.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.pigo_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.pigo_comm_uses_rccl.argtypes = [vp]
    L.pigo_comm_abort.argtypes = [vp]
    L.pigo_comm_destroy.argtypes = [vp]
    L.pigo_comm_destroy.restype = None
    L.pigo_shard_bounds.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.pigo_shard_bounds.restype = None
    L.pigo_wire_words.argtypes = [i32]
    L.pigo_wire_words.restype = sz
    L.pigo_wire_row_flags.argtypes = [vp]
    L.pigo_wire_row_flags.restype = C.c_int
    L.pigo_run_batch_sharded.argtypes = [vp, vp, vp, sz, i32, i32, dbl, i32, vp, vp]
    L.pigo_pack_lists.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.pigo_unpack_list.argtypes = [vp, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("pigo_device_count", "pigo_plan_last_timings", "pigo_wire_words", "pigo_wire_row_flags", "pigo_comm_uses_rccl"):
            fn.restype = C.c_int
    _lib = L
    return L


def check(status, what=""):
    if status == PIGO_OK:
        return
    msg = load_library().pigo_last_error().decode("utf-8", "replace")
    text = f"{what}: {msg}" if what else msg
    if status in (ERR_PACKET, ERR_PANIC):
        raise PigoPanic(text)  # the Go shim re-raises these as panics, like the reference
    if status == ERR_PARAM:
        raise ValueError(text)
    if status == ERR_NOMEM:
        raise MemoryError(text)
    raise PigoError(f"[{status}] {text}")


# ---- the reference's parameter structs (core/pigo.go:16-34) ---------------------------------------------------


@dataclass
class ImageParams:
    Pixels: np.ndarray = None  # row-major gray, stride Dim
    Rows: int = 0
    Cols: int = 0
    Dim: int = 0


@dataclass
class CascadeParams:
    MinSize: int = 0
    MaxSize: int = 0
    ShiftFactor: float = 0.0
    ScaleFactor: float = 0.0
    ImageParams: ImageParams = field(default_factory=ImageParams)


def make_dets(rows):
    """list of (row, col, scale, q) -> Detection array"""
    a = np.zeros(len(rows), dtype=DET_DTYPE)
    for i, r in enumerate(rows):
        a[i] = tuple(r)
    return a


def sort_by_q(dets):
    """sort.Slice(dets, func(i, j) bool { return dets[i].Q < dets[j].Q }) -- Go's pdqsort, in place."""
    assert dets.dtype == DET_DTYPE and dets.flags.c_contiguous
    load_library().pigo_sort_by_q(dets.ctypes.data, len(dets))
    return dets


class Pigo:
    """type Pigo (core/pigo.go:37-43).  NewPigo() == Pigo()."""

    def __init__(self, _handle=None, device=0):
        self._h = _handle
        self.device = device

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.pigo_cascade_destroy(h)

    def close(self):
        self.__del__()

    # Unpack returns a NEW *Pigo and leaves the receiver untouched (core/pigo.go:51,103-109)
    def Unpack(self, packet: bytes, device: int = None):
        L = load_library()
        dev = self.device if device is None else device
        h = C.c_void_p()
        check(L.pigo_cascade_create(bytes(packet), len(packet), dev, C.byref(h)), "Unpack")
        return Pigo(h, dev)

    def _need(self):
        if not self._h:
            raise PigoError("Pigo is not unpacked (call Unpack first)")
        return self._h

    @property
    def treeDepth(self):
        d, n = C.c_uint32(), C.c_uint32()
        check(load_library().pigo_cascade_info(self._need(), C.byref(d), C.byref(n)))
        return d.value

    @property
    def treeNum(self):
        d, n = C.c_uint32(), C.c_uint32()
        check(load_library().pigo_cascade_info(self._need(), C.byref(d), C.byref(n)))
        return n.value

    def tables(self):
        """(treeCodes int8[ntrees, 4*2^d], treePred f32[ntrees, 2^d], treeThreshold f32[ntrees])"""
        n, d = self.treeNum, self.treeDepth
        codes = np.zeros((n, 4 << d), dtype=np.int8)
        pred = np.zeros((n, 1 << d), dtype=np.float32)
        thr = np.zeros(n, dtype=np.float32)
        check(load_library().pigo_cascade_tables(self._need(), codes.ctypes.data, codes.size, pred.ctypes.data, pred.size,
                                                 thr.ctypes.data, thr.size))
        return codes, pred, thr

    # RunCascade, core/pigo.go:212
    def RunCascade(self, cp: CascadeParams, angle: float) -> np.ndarray:
        ip = cp.ImageParams
        pix = np.ascontiguousarray(ip.Pixels, dtype=np.uint8).ravel()
        L = load_library()
        cap = 1024
        while True:
            out = np.zeros(cap, dtype=DET_DTYPE)
            n = C.c_int(0)
            st = L.pigo_run_cascade(self._need(), pix.ctypes.data, pix.size, int(ip.Rows), int(ip.Cols), int(ip.Dim), int(cp.MinSize),
                                    int(cp.MaxSize), float(cp.ShiftFactor), float(cp.ScaleFactor), float(angle), out.ctypes.data, cap,
                                    C.byref(n))
            if st == ERR_CAPACITY and n.value > cap:
                cap = n.value
                continue
            check(st, "RunCascade")
            return out[: n.value].copy()

    # ClusterDetections, core/pigo.go:262 -- sorts `detections` in place, returns a fresh slice
    def ClusterDetections(self, detections: np.ndarray, iouThreshold: float) -> np.ndarray:
        assert detections.dtype == DET_DTYPE and detections.flags.c_contiguous
        n = len(detections)
        out = np.zeros(max(n, 1), dtype=DET_DTYPE)
        k = C.c_int(0)
        check(load_library().pigo_cluster_detections(self._need(), detections.ctypes.data, n, float(iouThreshold), out.ctypes.data,
                                                     len(out), C.byref(k)), "ClusterDetections")
        return out[: k.value].copy()


def NewPigo(device: int = 0) -> Pigo:
    """core/pigo.go:46"""
    return Pigo(device=device)


def RgbToGrayscale(src: np.ndarray, kind: int = PIX_NRGBA, device: int = 0) -> np.ndarray:
    """core/grayscale.go:8-23 -- ``src`` is the image's Pix as an (H, W, 4) uint8 array {R,G,B,A} (rows may be strided,
    like a Go sub-image); returns the width*height gray bytes RunCascade takes as ImageParams.Pixels.

    ``kind`` names the Go image type the reference would have been handed (PIX_NRGBA: what GetImage returns;
    PIX_RGBA: premultiplied) or PIX_CANVAS for the wasm front end's formula (wasm/canvas/canvas.go:179-191).
    """
    src = np.asarray(src)
    if src.dtype != np.uint8 or src.ndim != 3 or src.shape[2] != 4:
        raise ValueError("src must be an (H, W, 4) uint8 array")
    h, w = src.shape[:2]
    if h and w and (src.strides[2] != 1 or src.strides[1] != 4 or src.strides[0] < 4 * w):
        src = np.ascontiguousarray(src)
    out = np.zeros(h * w, dtype=np.uint8)
    if h == 0 or w == 0:
        return out
    stride = src.strides[0]
    npix = (h - 1) * stride + 4 * w
    check(load_library().pigo_rgb_to_grayscale(int(device), src.ctypes.data, npix, w, h, stride, int(kind), out.ctypes.data, out.size),
          "RgbToGrayscale")
    return out


# ---- PuplocCascade: pupil / facial-landmark localisation (core/puploc.go, core/flploc.go) ---------------------------------

#: Puploc wire record of the C ABI (pigo_puploc) and the batch request record (pigo_puploc_req)
PUPLOC_DTYPE = np.dtype([("row", "<i4"), ("col", "<i4"), ("scale", "<f4"), ("perturbs", "<i4")])
PUPLOC_REQ_DTYPE = np.dtype([("row", "<i4"), ("col", "<i4"), ("scale", "<f4"), ("perturbs", "<i4"), ("frame", "<i4"), ("flip_v", "<i4")])
POOL_SIZE = 63  # entries of each of the three sync.Pool arrays, puploc.go:231-235


@dataclass
class Puploc:
    """type Puploc, core/puploc.go:14-19"""
    Row: int = 0
    Col: int = 0
    Scale: float = 0.0
    Perturbs: int = 0


def new_pool() -> np.ndarray:
    """A brand-new sync.Pool object of RunDetector: rows | cols | scale, 63 float32 zeros each (puploc.go:228-237)."""
    return np.zeros((3, POOL_SIZE), dtype=np.float32)


def draw_perturbations(perturbs: int, rng=None) -> np.ndarray:
    """What RunDetector takes from rand.Float32(): 3 values per perturbation in draw order (puploc.go:248-250)."""
    rng = np.random.default_rng() if rng is None else rng
    return rng.random(3 * max(int(perturbs), 0), dtype=np.float32)


class PuplocCascade:
    """type PuplocCascade (core/puploc.go:22-29) with device-resident tables.  NewPuplocCascade() == PuplocCascade()."""

    def __init__(self, _handle=None, device=0):
        self._h = _handle
        self.device = device

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.pigo_puploc_destroy(h)

    def _need(self):
        if not self._h:
            raise PigoError("PuplocCascade is not unpacked (call UnpackCascade first)")
        return self._h

    # UnpackCascade, core/puploc.go:38-103: returns a NEW cascade
    def UnpackCascade(self, packet: bytes, device: int = None):
        dev = self.device if device is None else device
        h = C.c_void_p()
        check(load_library().pigo_puploc_create(bytes(packet), len(packet), dev, C.byref(h)), "UnpackCascade")
        return PuplocCascade(h, dev)

    # UnpackFlp, core/flploc.go:27-33
    def UnpackFlp(self, cf: str):
        with open(cf, "rb") as fh:
            return self.UnpackCascade(fh.read())

    # ReadCascadeDir, core/flploc.go:60-81: file name -> [cascade]
    def ReadCascadeDir(self, path: str):
        names = sorted(os.listdir(path))
        if not names:
            raise PigoError("the provided directory is empty")
        return {n: [self.UnpackFlp(os.path.join(os.path.abspath(path), n))] for n in names}

    @property
    def header(self):
        """(stages, scales, trees, treeDepth)"""
        st, tr, d, sc = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_float()
        check(load_library().pigo_puploc_info(self._need(), C.byref(st), C.byref(sc), C.byref(tr), C.byref(d)))
        return st.value, sc.value, tr.value, d.value

    @staticmethod
    def _args(img: ImageParams, perturbs, rnd, pool):
        pix = np.ascontiguousarray(img.Pixels, dtype=np.uint8).ravel()
        rnd = draw_perturbations(perturbs) if rnd is None else np.ascontiguousarray(rnd, dtype=np.float32).ravel()
        if rnd.size < 3 * min(max(int(perturbs), 0), POOL_SIZE):
            raise ValueError("rnd needs 3 values per perturbation")
        if pool is not None and not (pool.dtype == np.float32 and pool.shape == (3, POOL_SIZE) and pool.flags.c_contiguous):
            raise ValueError("pool must be a contiguous float32 [3, 63] array (see new_pool())")
        return pix, rnd, (pool.ctypes.data if pool is not None else None)

    # RunDetector, core/puploc.go:239-277
    def RunDetector(self, pl: Puploc, img: ImageParams, angle: float, flipV: bool, rnd=None, pool=None) -> Puploc:
        """``rnd``: the 3*Perturbs values rand.Float32() would return (None: drawn here, like the reference's global
        source); ``pool``: the sync.Pool object, float32 [3, 63] read and written (None: a brand-new one)."""
        pix, rnd, pp = self._args(img, pl.Perturbs, rnd, pool)
        req = np.zeros(1, dtype=PUPLOC_DTYPE)
        req[0] = (int(pl.Row), int(pl.Col), float(pl.Scale), int(pl.Perturbs))
        out = np.zeros(1, dtype=PUPLOC_DTYPE)
        check(load_library().pigo_puploc_run_detector(self._need(), req.ctypes.data, pix.ctypes.data, pix.size, int(img.Rows), int(img.Cols),
                                                      int(img.Dim), float(angle), int(bool(flipV)), rnd.ctypes.data, pp, out.ctypes.data),
              "RunDetector")
        return Puploc(int(out[0]["row"]), int(out[0]["col"]), float(out[0]["scale"]), 0)

    # GetLandmarkPoint, core/flploc.go:36-57
    def GetLandmarkPoint(self, leftEye: Puploc, rightEye: Puploc, img: ImageParams, perturb: int, flipV: bool, rnd=None, pool=None) -> Puploc:
        pix, rnd, pp = self._args(img, perturb, rnd, pool)
        eyes = np.zeros(2, dtype=PUPLOC_DTYPE)
        eyes[0] = (int(leftEye.Row), int(leftEye.Col), float(leftEye.Scale), 0)
        eyes[1] = (int(rightEye.Row), int(rightEye.Col), float(rightEye.Scale), 0)
        out = np.zeros(1, dtype=PUPLOC_DTYPE)
        check(load_library().pigo_get_landmark_point(self._need(), eyes[0:1].ctypes.data, eyes[1:2].ctypes.data, pix.ctypes.data, pix.size,
                                                     int(img.Rows), int(img.Cols), int(img.Dim), int(perturb), int(bool(flipV)),
                                                     rnd.ctypes.data, pp, out.ctypes.data), "GetLandmarkPoint")
        return Puploc(int(out[0]["row"]), int(out[0]["col"]), float(out[0]["scale"]), 0)


def NewPuplocCascade(device: int = 0) -> PuplocCascade:
    """core/puploc.go:32-34"""
    return PuplocCascade(device=device)
