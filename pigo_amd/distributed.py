"""Multi-GPU batch scan: one process per GPU, frames sharded, detection lists all-gathered.

Frames are independent (RunCascade has no cross-frame state, core/pigo.go:212-258), so the batch is
split into contiguous shards with no exchange during the scan; the only collective is ONE all-gather of
fixed-capacity per-frame detection records at the end (RCCL over xGMI when the backend is "nccl").
RCCL has no all-gather-v, so each frame contributes ``2 + 4*gather_cap`` int32 words: the true count, a
flags word (WIRE_* below: truncated at the gather capacity / at det_cap, the producing rank's queue overflow,
would-panic, rank failed, padding row -- what a peer must know without asking the rank that made the row),
then ``gather_cap`` 16-byte records (row, col, scale, q-bits).

Two ways to run it, same wire format:
  * ``Comm`` + ``run_batch_sharded`` -- the C ABI (include/pigo_hip.h: pigo_comm_init / pigo_run_batch_sharded): scan,
    cluster, pack and ONE ``ncclAllGather`` issued by libpigo_hip.so itself on the caller's stream.  This is what a Go or
    C++ host uses and what bench.py times; only the 128-byte RCCL id travels through the host program's own channel
    (here: ``torch.distributed``'s store).
  * ``allgather_lists`` -- the same rows through ``torch.distributed.all_gather_into_tensor`` (RCCL when the backend is
    "nccl"); backend-agnostic, it also runs on CPU tensors with gloo, which is how the tests cover N > 1 without GPUs.
"""
import ctypes as C

import numpy as np

from . import core


def shard_bounds(nframes: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `nframes` frames for `rank`; earlier ranks take the remainder (== pigo_shard_bounds)."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


#: flags word of a wire row (include/pigo_hip.h PIGO_WIRE_*)
WIRE_TRUNCATED_GATHER, WIRE_TRUNCATED_DETCAP, WIRE_QUEUE_OVERFLOW, WIRE_WOULD_PANIC, WIRE_RANK_FAILED, WIRE_PADDING = 1, 2, 4, 8, 16, 32
WIRE_HEAD = 2  # words in front of a row's records: the true count, the flags


def pack_lists(dets, counts, gather_cap: int, raw_counts=None, plan_flags=None):
    """dets int32 [n, cap, 4], counts int32 [n] -> int32 [n, 2 + 4*gather_cap] wire rows: the true count, the flags word, then
    the first min(count, gather_cap) records, zero-padded (the layout k_pack_lists / pigo_pack_lists produce).  raw_counts: the
    RunCascade counts a clustered list was made from; plan_flags: the (queue, panic) flags of the scan, as last_flags() gives them."""
    import torch
    n, cap = dets.shape[0], dets.shape[1]
    if gather_cap > cap:
        raise ValueError(f"gather_cap {gather_cap} exceeds the lists' capacity {cap}")
    g = gather_cap
    wire = torch.zeros((n, WIRE_HEAD + 4 * gather_cap), dtype=torch.int32, device=dets.device)
    wire[:, 0] = counts
    cut = counts > cap
    if raw_counts is not None:
        cut = cut | (raw_counts > cap)
    flags = (counts > g).to(torch.int32) * WIRE_TRUNCATED_GATHER + cut.to(torch.int32) * WIRE_TRUNCATED_DETCAP
    if plan_flags is not None:
        flags = flags + (WIRE_QUEUE_OVERFLOW if plan_flags[0] else 0) + (WIRE_WOULD_PANIC if plan_flags[1] else 0)
    wire[:, 1] = flags
    keep = torch.arange(g, device=dets.device).unsqueeze(0) < counts.clamp(max=g).unsqueeze(1)  # [n, g]
    wire[:, WIRE_HEAD:WIRE_HEAD + 4 * g] = (dets[:, :g, :] * keep.unsqueeze(-1).to(dets.dtype)).reshape(n, 4 * g)
    return wire


class Comm:
    """pigo_comm (include/pigo_hip.h): this rank's handle on the RCCL communicator libpigo_hip.so uses for the all-gather.

    ``Comm.from_torch(device)`` builds it inside an initialised ``torch.distributed`` job: rank 0 asks the library for the
    RCCL id and the process group only carries those 128 bytes to the other ranks."""

    def __init__(self, rank: int, world: int, device: int, unique_id: bytes = None):
        self.L = core.load_library()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        h = C.c_void_p()
        idbuf = (C.c_uint8 * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        core.check(self.L.pigo_comm_init(idbuf, self.rank, self.world, self.device, C.byref(h)), "comm_init")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and core._lib is not None:
            core._lib.pigo_comm_destroy(h)

    def abort(self):
        """pigo_comm_abort (ncclCommAbort): give the communicator up without waiting for outstanding collectives -- a peer failed."""
        core.check(self.L.pigo_comm_abort(self._h), "comm_abort")

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        core.check(core.load_library().pigo_comm_unique_id(buf), "comm_unique_id")
        return bytes(buf)

    @property
    def uses_rccl(self) -> bool:
        """True when the all-gather of run_batch_sharded is ncclAllGather (world > 1, or world == 1 built with an id)."""
        return bool(self.L.pigo_comm_uses_rccl(self._h))

    @classmethod
    def from_torch(cls, device: int, group=None, force_rccl: bool = False):
        """``force_rccl``: at world size 1 build a real one-rank RCCL communicator as well (ncclGetUniqueId ->
        ncclCommInitRank -> ncclAllGather), so that a single-GPU box runs the very collective an 8-GPU node does."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 and (world > 1 or force_rccl) else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return cls(rank, world, device, box[0])


def run_batch_sharded(plan, comm, frames, frames_per_rank: int, iou: float, gather_cap: int, out=None, stream=None):
    """pigo_run_batch_sharded: scan this rank's ``frames`` (uint8 [n, rows, dim] on the GPU, n <= frames_per_rank),
    cluster per frame (iou >= 0; a negative iou gathers the raw lists) and all-gather the wire rows of all ranks.
    Returns the int32 [world*frames_per_rank, 2 + 4*gather_cap] tensor (``out`` to reuse it).  Asynchronous on the current
    stream; ``plan.status()`` applies after synchronising."""
    import torch
    from .batch import ScanPlan
    n, stride = plan._check_frames(frames) if frames is not None and frames.shape[0] else (0, plan.rows * plan.dim)
    world = comm.world if comm is not None else 1
    words = WIRE_HEAD + 4 * int(gather_cap)
    if out is None:
        out = torch.zeros((world * frames_per_rank, words), dtype=torch.int32, device=torch.device("cuda", plan.device))
    assert out.dtype == torch.int32 and out.is_cuda and out.is_contiguous() and tuple(out.shape) == (world * frames_per_rank, words)
    core.check(plan.L.pigo_run_batch_sharded(plan._h, comm._h if comm is not None else None,
                                             C.c_void_p(frames.data_ptr()) if n else None, stride, n, int(frames_per_rank), float(iou),
                                             int(gather_cap), C.c_void_p(out.data_ptr()), ScanPlan._stream_ptr(stream)), "run_batch_sharded")
    return out


def pack_lists_host(dets: np.ndarray, counts: np.ndarray, frames_out: int, gather_cap: int) -> np.ndarray:
    """pigo_pack_lists on host arrays: dets DET_DTYPE [n, cap], counts int32 [n] -> int32 [frames_out, 2 + 4*gather_cap]."""
    dets = np.ascontiguousarray(dets)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n, cap = dets.shape
    wire = np.zeros((frames_out, WIRE_HEAD + 4 * gather_cap), dtype=np.int32)
    core.check(core.load_library().pigo_pack_lists(dets.ctypes.data, counts.ctypes.data, n, frames_out, cap, gather_cap, wire.ctypes.data),
               "pack_lists")
    return wire


def unpack_list_host(row: np.ndarray, gather_cap: int):
    """pigo_unpack_list: one wire row -> (Detection array, true count)."""
    row = np.ascontiguousarray(row, dtype=np.int32)
    out = np.zeros(max(gather_cap, 1), dtype=core.DET_DTYPE)
    n, cnt = C.c_int(0), C.c_int(0)
    core.check(core.load_library().pigo_unpack_list(row.ctypes.data, gather_cap, out.ctypes.data, len(out), C.byref(n), C.byref(cnt)),
               "unpack_list")
    return out[: n.value].copy(), cnt.value


def unpack_lists(wire, gather_cap: int):
    """Inverse of pack_lists on the host: list of Detection arrays + the true counts."""
    w = wire.cpu().numpy()
    counts = w[:, 0].copy()
    out = []
    for f in range(w.shape[0]):
        n = min(int(counts[f]), gather_cap)
        out.append(np.ascontiguousarray(w[f, WIRE_HEAD:WIRE_HEAD + 4 * n]).view(core.DET_DTYPE).reshape(n).copy())
    return out, counts


def row_flags(wire):
    """The flags words of gathered rows (tensor or array [rows, words]) as a NumPy array: WIRE_* bits."""
    w = wire.cpu().numpy() if hasattr(wire, "cpu") else np.asarray(wire)
    return w[:, 1].copy()


def allgather_lists(dets, counts, gather_cap: int, frames_per_rank: int, group=None, raw_counts=None, plan_flags=None, rank_failed=False):
    """All-gather every rank's per-frame lists over ``torch.distributed``.  Every rank must pass exactly `frames_per_rank` frames'
    worth of rows (pad short shards with zero-count frames).  Returns int32 [world*frames_per_rank, 2+4*gather_cap].

    The rows carry the same flags word as the C ABI's (pigo_run_batch_sharded packs it on the device): pass ``raw_counts`` (the
    RunCascade counts a clustered list was made from: TRUNCATED_DETCAP when the RAW list was cut), ``plan_flags`` (``plan.last_flags()``
    after the scan has been synchronised: QUEUE_OVERFLOW, WOULD_PANIC) and ``rank_failed=True`` when this rank's scan was refused
    and its rows are padding -- a peer then reads from the rows what it would read from the C ABI's.  Without them the flags word
    only says what the lists themselves show (cut at gather_cap / at det_cap)."""
    import torch
    import torch.distributed as dist
    wire = pack_lists(dets, counts, gather_cap, raw_counts=raw_counts, plan_flags=plan_flags)
    if rank_failed:
        wire = torch.zeros_like(wire)
        wire[:, 1] = WIRE_RANK_FAILED
    if wire.shape[0] < frames_per_rank:
        pad = torch.zeros((frames_per_rank - wire.shape[0], wire.shape[1]), dtype=wire.dtype, device=wire.device)
        pad[:, 1] = WIRE_PADDING
        wire = torch.cat([wire, pad], dim=0)
    world = dist.get_world_size(group)
    out = torch.empty((world * frames_per_rank, wire.shape[1]), dtype=wire.dtype, device=wire.device)
    dist.all_gather_into_tensor(out, wire.contiguous(), group=group)
    return out


def gathered_frame_index(nframes: int, world: int):
    """Row r of the gathered tensor -> global frame index (or -1 for padding rows)."""
    per = (nframes + world - 1) // world
    idx = np.full(world * per, -1, dtype=np.int64)
    for rank in range(world):
        lo, hi = shard_bounds(nframes, rank, world)
        idx[rank * per: rank * per + (hi - lo)] = np.arange(lo, hi)
    return idx, per
