"""world_size-2 test of the N > 1 path on CPU (gloo): shard -> (scan stand-in) -> pack -> all-gather -> unpack.
The scan itself needs a GPU; here each rank fabricates its shard's detection lists deterministically, which
exercises exactly the code bench.py and the sharded API run around the kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pigo_amd import core, distributed


def _fake_lists(frame_lo, frame_hi, cap):
    n = frame_hi - frame_lo
    dets = torch.zeros((n, cap, 4), dtype=torch.int32)
    counts = torch.zeros((n,), dtype=torch.int32)
    for i, f in enumerate(range(frame_lo, frame_hi)):
        k = (f * 7) % (cap + 3)  # some frames exceed the gather capacity
        counts[i] = k
        for j in range(min(k, cap)):
            q = np.float32(f + j / 10.0)
            dets[i, j] = torch.tensor([f, j, 20 + j, int(q.view(np.int32))], dtype=torch.int32)
    return dets, counts


def _worker(rank, world, port, nframes, cap, gcap, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = distributed.shard_bounds(nframes, rank, world)
        dets, counts = _fake_lists(lo, hi, cap)
        idx, per = distributed.gathered_frame_index(nframes, world)
        wire = distributed.allgather_lists(dets, counts, gcap, per)
        lists, cnt = distributed.unpack_lists(wire, gcap)
        ok = wire.shape[0] == per * world
        for row, f in enumerate(idx.tolist()):
            if f < 0:
                ok &= cnt[row] == 0 and len(lists[row]) == 0
                continue
            k = (f * 7) % (cap + 3)
            ok &= int(cnt[row]) == k and len(lists[row]) == min(k, gcap, cap)
            for j, d in enumerate(lists[row]):
                ok &= (int(d["row"]), int(d["col"]), int(d["scale"])) == (f, j, 20 + j) and d["q"] == np.float32(f + j / 10.0)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_allgather_lists_gloo_world2():
    world, nframes, cap, gcap = 2, 7, 12, 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), nframes, cap, gcap, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_pack_unpack_roundtrip():
    dets, counts = _fake_lists(0, 5, 12)
    wire = distributed.pack_lists(dets, counts, 8)
    lists, cnt = distributed.unpack_lists(wire, 8)
    assert cnt.tolist() == counts.tolist()  # the true counts survive, even when > gather_cap
    for f, l in enumerate(lists):
        assert len(l) == min(int(counts[f]), 8) and l.dtype == core.DET_DTYPE
    import pytest
    with pytest.raises(ValueError):
        distributed.pack_lists(dets, counts, 16)  # gather_cap larger than the lists' capacity
