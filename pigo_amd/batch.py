"""Device-resident batch scan: the extension of RunCascade that BASELINE configs 2-5 measure.

A ``ScanPlan`` fixes CascadeParams (minus the pixels) and owns the GPU workspace for a batch of frames
that already live in HBM.  PyTorch is used only as plumbing here -- device memory (uint8 / int32 tensors
whose ``data_ptr()`` is handed to the C ABI), the current HIP stream and ``torch.distributed`` -- all the
work happens in libpigo_hip.so.
"""
import ctypes as C

import numpy as np

from . import core


def _torch():
    import torch
    return torch


class ScanPlan:
    """pigo_plan (include/pigo_hip.h): CascadeParams bound to a cascade + workspace for `max_frames`."""

    def __init__(self, pigo: "core.Pigo", rows, cols, dim=None, MinSize=20, MaxSize=1000, ShiftFactor=0.1, ScaleFactor=1.1,
                 angle=0.0, max_frames=1, det_cap=4096):
        self.L = core.load_library()
        self.pigo = pigo
        self.rows, self.cols, self.dim = int(rows), int(cols), int(dim if dim is not None else cols)
        self.max_frames, self.det_cap = int(max_frames), int(det_cap)
        h = C.c_void_p()
        core.check(self.L.pigo_plan_create(pigo._need(), self.rows, self.cols, self.dim, int(MinSize), int(MaxSize), float(ShiftFactor),
                                           float(ScaleFactor), float(angle), self.max_frames, self.det_cap, C.byref(h)), "plan_create")
        self._h = h
        self.device = pigo.device

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and core is not None and core._lib is not None:
            core._lib.pigo_plan_destroy(h)

    def info(self) -> core.PlanInfo:
        inf = core.PlanInfo()
        core.check(self.L.pigo_plan_info(self._h, C.byref(inf)))
        return inf

    def set_variant(self, variant: int):
        core.check(self.L.pigo_plan_set_variant(self._h, int(variant)), "set_variant")

    def set_profiling(self, on: bool):
        core.check(self.L.pigo_plan_set_profiling(self._h, 1 if on else 0))

    # -- buffers ---------------------------------------------------------------------------------------------
    def alloc_outputs(self, nframes=None):
        """(dets uint8-viewed-as-records [n, det_cap, 16 B], counts int32 [n]) on the plan's device."""
        torch = _torch()
        n = self.max_frames if nframes is None else nframes
        dev = torch.device("cuda", self.device)
        dets = torch.zeros((n, self.det_cap, 4), dtype=torch.int32, device=dev)
        counts = torch.zeros((n,), dtype=torch.int32, device=dev)
        return dets, counts

    @staticmethod
    def _stream_ptr(stream):
        torch = _torch()
        s = torch.cuda.current_stream() if stream is None else stream
        return C.c_void_p(s.cuda_stream)

    def _check_frames(self, frames):
        torch = _torch()
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.is_contiguous()
        assert frames.dim() == 3 and frames.shape[1] == self.rows and frames.shape[2] == self.dim, frames.shape
        assert frames.shape[0] <= self.max_frames
        return int(frames.shape[0]), self.rows * self.dim

    # -- RunCascade over a batch --------------------------------------------------------------------------------
    def run(self, frames, dets, counts, stream=None, sync=False):
        """Enqueue the scan of ``frames`` (uint8 [n, rows, dim], device) on the current stream.

        dets: int32 [n, det_cap, 4] (row, col, scale, q-bits) in the reference's order; counts: int32 [n]."""
        n, stride = self._check_frames(frames)
        fn = self.L.pigo_plan_run_sync if sync else self.L.pigo_plan_run
        core.check(fn(self._h, C.c_void_p(frames.data_ptr()), stride, n, C.c_void_p(dets.data_ptr()), C.c_void_p(counts.data_ptr()),
                      self._stream_ptr(stream)), "plan_run")

    def status(self):
        """Call after synchronising: raises PigoPanic / PigoError(capacity) like the single-frame API."""
        core.check(self.L.pigo_plan_status(self._h), "plan_status")

    def last_flags(self):
        """(queue_overflow, would_panic, det_cap_overflow) as the last ``status()`` read them (pigo_plan_last_flags): tells
        'results incomplete, re-run' (queue) from 'list truncated, re-plan with a larger det_cap'."""
        q, pn, dc = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        core.check(self.L.pigo_plan_last_flags(self._h, C.byref(q), C.byref(pn), C.byref(dc)), "plan_last_flags")
        return q.value, pn.value, dc.value

    def alloc_cluster_outputs(self, dets, counts):
        torch = _torch()
        return torch.zeros_like(dets), torch.zeros_like(dets), torch.zeros_like(counts), torch.zeros_like(counts)

    def cluster(self, dets, counts, iou, stream=None, out=None):
        """Per-frame ClusterDetections on the GPU.  Returns (sorted, clusters, ccounts, ties) tensors
        (pass ``out=alloc_cluster_outputs(...)`` to reuse buffers)."""
        n = int(counts.shape[0])
        sorted_, clusters, ccounts, ties = out if out is not None else self.alloc_cluster_outputs(dets, counts)
        core.check(self.L.pigo_plan_cluster(self._h, C.c_void_p(dets.data_ptr()), C.c_void_p(counts.data_ptr()), n, float(iou),
                                            C.c_void_p(sorted_.data_ptr()), C.c_void_p(clusters.data_ptr()),
                                            C.c_void_p(ccounts.data_ptr()), C.c_void_p(ties.data_ptr()), self._stream_ptr(stream)),
                   "plan_cluster")
        return sorted_, clusters, ccounts, ties

    def last_timings(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        k = self.L.pigo_plan_last_timings(self._h, names, ms, 16)
        return [(names[i].decode(), float(ms[i])) for i in range(k)]

    def debug_stats(self):
        out = (C.c_uint64 * 16)()
        core.check(self.L.pigo_plan_debug_stats(self._h, out, 16))
        return [int(v) for v in out]

    def debug_trace(self):
        """Debug build + PIGO_DEBUG_STATS=1: the 16 x 256 trace words of the last run (k_scan_one: per workgroup
        {start, first item done, all items done, wave 0 leaves} in 10-ns ticks, items, first item, entries taken)."""
        out = (C.c_uint64 * (16 * 256))()
        core.check(self.L.pigo_plan_debug_trace(self._h, out, 16 * 256))
        return np.frombuffer(out, dtype=np.uint64).reshape(256, 16).copy()

    def last_queue_count(self):
        n = C.c_int64(0)
        core.check(self.L.pigo_plan_last_queue_count(self._h, C.byref(n)))
        return n.value


def dets_to_numpy(dets, counts, frame=None):
    """Device detection tensors -> list of Detection arrays (one per frame), truncated to det_cap."""
    d = dets.cpu().numpy()
    c = counts.cpu().numpy()
    cap = d.shape[1]
    out = []
    for f in range(d.shape[0]):
        n = min(int(c[f]), cap)
        out.append(np.ascontiguousarray(d[f, :n]).view(core.DET_DTYPE).reshape(n).copy())
    return out if frame is None else out[frame]


def rgb_to_grayscale(rgba, kind=core.PIX_NRGBA, out=None, dim=None, stream=None):
    """RgbToGrayscale (core/grayscale.go:8-23) over a batch of device-resident frames.

    rgba: uint8 [n, rows, cols, 4] {R,G,B,A} on the GPU (contiguous); returns uint8 [n, rows, dim] gray frames
    (dim >= cols, default cols) that ``ScanPlan.run`` takes as is.  Enqueued on the current stream, nothing synchronised.
    """
    torch = _torch()
    assert rgba.dtype == torch.uint8 and rgba.is_cuda and rgba.is_contiguous() and rgba.dim() == 4 and rgba.shape[3] == 4, rgba.shape
    n, rows, cols = (int(v) for v in rgba.shape[:3])
    dim = cols if dim is None else int(dim)
    if out is None:
        out = torch.zeros((n, rows, dim), dtype=torch.uint8, device=rgba.device)
    assert out.dtype == torch.uint8 and out.is_cuda and out.is_contiguous() and tuple(out.shape) == (n, rows, dim), out.shape
    L = core.load_library()
    dev = rgba.device.index if rgba.device.index is not None else torch.cuda.current_device()
    core.check(L.pigo_gray_batch(dev, C.c_void_p(rgba.data_ptr()), rows * cols * 4, cols * 4, cols, rows, int(kind), n,
                                 C.c_void_p(out.data_ptr()), rows * dim, dim, ScanPlan._stream_ptr(stream)), "gray_batch")
    return out


def puploc_run_batch(plc: "core.PuplocCascade", frames, reqs, rnd, angle=0.0, pool=None, out=None, stream=None, cols=None):
    """n independent RunDetector calls (core/puploc.go:239-277) against device-resident gray frames, one launch.

    frames: uint8 [f, rows, dim] on the GPU, of which the first `cols` (default dim) columns of every row are image --
    ImageParams.Cols, what the reference clamps column indices with (puploc.go:118-128); reqs: uint8/int32 tensor holding n ``core.PUPLOC_REQ_DTYPE`` records
    (24 B each: row, col, scale, perturbs, frame, flip_v); rnd: float32 [n, 189] perturbation randoms in draw order;
    pool: float32 [n, 189] in/out or None (fresh pool objects).  Returns an int32 [n, 4] tensor of ``core.PUPLOC_DTYPE``
    records (row, col, scale-bits, 0).  Call ``puploc_status(plc)`` after synchronising."""
    torch = _torch()
    assert frames.dtype == torch.uint8 and frames.is_cuda and frames.is_contiguous() and frames.dim() == 3
    nf, rows, dim = (int(v) for v in frames.shape)
    cols = dim if cols is None else int(cols)
    assert 1 <= cols <= dim, (cols, dim)
    assert reqs.is_cuda and reqs.is_contiguous() and reqs.numel() * reqs.element_size() % 24 == 0
    n = reqs.numel() * reqs.element_size() // 24
    assert rnd.dtype == torch.float32 and rnd.is_cuda and rnd.is_contiguous() and tuple(rnd.shape) == (n, 3 * core.POOL_SIZE), rnd.shape
    if pool is not None:
        assert pool.dtype == torch.float32 and pool.is_cuda and pool.is_contiguous() and tuple(pool.shape) == (n, 3 * core.POOL_SIZE)
    if out is None:
        out = torch.zeros((n, 4), dtype=torch.int32, device=frames.device)
    L = core.load_library()
    core.check(L.pigo_puploc_run_batch(plc._need(), C.c_void_p(frames.data_ptr()), rows * dim, nf, rows, cols, dim, float(angle),
                                       C.c_void_p(reqs.data_ptr()), C.c_void_p(rnd.data_ptr()),
                                       C.c_void_p(pool.data_ptr()) if pool is not None else None, n, C.c_void_p(out.data_ptr()),
                                       ScanPlan._stream_ptr(stream)), "puploc_run_batch")
    return out


def puploc_status(plc: "core.PuplocCascade"):
    core.check(core.load_library().pigo_puploc_status(plc._need()), "puploc_status")
