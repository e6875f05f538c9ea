"""Multi-GPU batch scan: one process per GPU, frames sharded, detection lists all-gathered.

Frames are independent (RunCascade has no cross-frame state, core/pigo.go:212-258), so the batch is
split into contiguous shards with no exchange during the scan; the only collective is ONE all-gather of
fixed-capacity per-frame detection records at the end (RCCL over xGMI when the backend is "nccl").
RCCL has no all-gather-v, so each frame contributes ``1 + 4*gather_cap`` int32 words: the true count
followed by ``gather_cap`` 16-byte records (row, col, scale, q-bits); frames with more detections than
``gather_cap`` are visible as ``count > gather_cap``.

The packing / gathering code is backend-agnostic (works on CPU tensors with gloo), which is how the
tests exercise the N > 1 path without GPUs.
"""
import numpy as np

from . import core


def shard_bounds(nframes: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `nframes` frames for `rank`; earlier ranks take the remainder."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_lists(dets, counts, gather_cap: int):
    """dets int32 [n, cap, 4], counts int32 [n] -> int32 [n, 1 + 4*gather_cap] wire records."""
    import torch
    n, cap = dets.shape[0], dets.shape[1]
    if gather_cap > cap:
        raise ValueError(f"gather_cap {gather_cap} exceeds the lists' capacity {cap}")
    g = gather_cap
    wire = torch.zeros((n, 1 + 4 * gather_cap), dtype=torch.int32, device=dets.device)
    wire[:, 0] = counts
    wire[:, 1:1 + 4 * g] = dets[:, :g, :].reshape(n, 4 * g)
    return wire


def unpack_lists(wire, gather_cap: int):
    """Inverse of pack_lists on the host: list of Detection arrays + the true counts."""
    w = wire.cpu().numpy()
    counts = w[:, 0].copy()
    out = []
    for f in range(w.shape[0]):
        n = min(int(counts[f]), gather_cap)
        out.append(np.ascontiguousarray(w[f, 1:1 + 4 * n]).view(core.DET_DTYPE).reshape(n).copy())
    return out, counts


def allgather_lists(dets, counts, gather_cap: int, frames_per_rank: int, group=None):
    """All-gather every rank's per-frame lists.  Every rank must pass exactly `frames_per_rank` frames'
    worth of rows (pad short shards with zero-count frames).  Returns int32 [world*frames_per_rank, 1+4*gather_cap]."""
    import torch
    import torch.distributed as dist
    wire = pack_lists(dets, counts, gather_cap)
    if wire.shape[0] < frames_per_rank:
        pad = torch.zeros((frames_per_rank - wire.shape[0], wire.shape[1]), dtype=wire.dtype, device=wire.device)
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
