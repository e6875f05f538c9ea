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


def _worker_cabi(rank, world, port, nframes, cap, gcap, ret):
    """The same exchange with the C ABI's host-side wire functions (pigo_shard_bounds / pigo_pack_lists /
    pigo_unpack_list, include/pigo_hip.h) doing what k_pack_lists does on the device; gloo stands in for ncclAllGather."""
    import ctypes as C
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = core.load_library()
        lo, hi = C.c_int(0), C.c_int(0)
        L.pigo_shard_bounds(nframes, rank, world, C.byref(lo), C.byref(hi))
        ok = (lo.value, hi.value) == distributed.shard_bounds(nframes, rank, world)
        dets, counts = _fake_lists(lo.value, hi.value, cap)
        idx, per = distributed.gathered_frame_index(nframes, world)
        wire = distributed.pack_lists_host(dets.numpy().view(core.DET_DTYPE).reshape(hi.value - lo.value, cap), counts.numpy(), per, gcap)
        ok &= wire.shape == (per, int(L.pigo_wire_words(gcap)))
        ref = distributed.pack_lists(dets, counts, gcap).numpy()  # the torch path writes the same rows
        nloc = hi.value - lo.value
        ok &= bool((wire[:nloc] == ref).all())
        # padding rows: count 0, no records, and the flags word says so (a frame WITHOUT detections has flags 0)
        ok &= bool((wire[nloc:, 0] == 0).all()) and bool((wire[nloc:, 1] == distributed.WIRE_PADDING).all()) and bool((wire[nloc:, 2:] == 0).all())
        ok &= all(int(L.pigo_wire_row_flags(wire[r].ctypes.data)) == int(wire[r, 1]) for r in range(per))
        out = torch.empty((world * per, wire.shape[1]), dtype=torch.int32)
        dist.all_gather_into_tensor(out, torch.from_numpy(wire))
        for row, f in enumerate(idx.tolist()):
            lst, cnt = distributed.unpack_list_host(out[row].numpy(), gcap)
            if f < 0:
                ok &= cnt == 0 and len(lst) == 0
                continue
            k = (f * 7) % (cap + 3)
            ok &= cnt == k and len(lst) == min(k, gcap, cap)
            for j, d in enumerate(lst):
                ok &= (int(d["row"]), int(d["col"]), int(d["scale"])) == (f, j, 20 + j) and d["q"] == np.float32(f + j / 10.0)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _worker_flags(rank, world, port, nframes, cap, gcap, ret):
    """allgather_lists with the scan's flags handed through: the torch.distributed path writes the flags word the C ABI's
    device-side packer writes (raw lists cut at det_cap, the producing rank's queue / would-panic flags, a rank whose scan failed)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = distributed.shard_bounds(nframes, rank, world)
        dets, counts = _fake_lists(lo, hi, cap)
        idx, per = distributed.gathered_frame_index(nframes, world)
        raw = counts.clone()
        if rank == 0:
            raw[1] = cap + 5  # the RAW list of rank 0's second frame was cut at det_cap
        wire = distributed.allgather_lists(dets, counts, gcap, per, raw_counts=raw, plan_flags=(1, 0, 0) if rank == 1 else (0, 0, 0))
        fl = distributed.row_flags(wire)
        ok = True
        for row, f in enumerate(idx.tolist()):
            if f < 0:
                ok &= int(fl[row]) == distributed.WIRE_PADDING
                continue
            k = (f * 7) % (cap + 3)
            want = (distributed.WIRE_TRUNCATED_GATHER if k > gcap else 0) | (distributed.WIRE_TRUNCATED_DETCAP if (k > cap or f == 1) else 0)
            if row >= per:
                want |= distributed.WIRE_QUEUE_OVERFLOW  # every row of rank 1 says its queue overflowed
            ok &= int(fl[row]) == want
        # a rank whose scan was refused: padding rows that say so
        wire2 = distributed.allgather_lists(dets, counts, gcap, per, rank_failed=(rank == 1))
        fl2, cnt2 = distributed.row_flags(wire2), wire2[:, 0].numpy()
        nloc1 = distributed.shard_bounds(nframes, 1, world)
        n1 = nloc1[1] - nloc1[0]
        ok &= bool((fl2[per:per + n1] == distributed.WIRE_RANK_FAILED).all()) and bool((cnt2[per:] == 0).all())
        ok &= bool((fl2[:per] & distributed.WIRE_RANK_FAILED == 0).all())
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_allgather_lists_carries_the_scan_flags_gloo_world2():
    world, nframes, cap, gcap = 2, 7, 12, 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_flags, args=(world, _free_port(), nframes, cap, gcap, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


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


def test_c_abi_wire_format_gloo_world2():
    world, nframes, cap, gcap = 2, 7, 12, 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_cabi, args=(world, _free_port(), nframes, cap, gcap, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_comm_world1_needs_no_rccl_and_no_gpu():
    """pigo_comm_init(world = 1) is a plain handle: nothing to exchange, RCCL is not even loaded."""
    import pytest
    c = distributed.Comm(0, 1, 0)
    import ctypes as C
    r, w = C.c_int(-1), C.c_int(-1)
    core.check(core.load_library().pigo_comm_info(c._h, C.byref(r), C.byref(w)))
    assert (r.value, w.value) == (0, 1)
    assert c.uses_rccl is False  # (a one-rank communicator built WITH an id goes through RCCL: tests/test_gpu_parity.py)
    with pytest.raises(ValueError):
        distributed.Comm(2, 2, 0, bytes(128))  # rank outside [0, world)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself (torch.distributed.run) and report
    n_gpus 2 with the all-gather inside the step; --cpu-dry-run swaps the GPU scan for fabricated lists and RCCL for gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-dry-run", "--steps", "2", "--warmup", "1",
                        "--frames", "5"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["ranks_seen"] == [0, 1] and d["gathered_rows"] == 10


def test_bench_dry_run_world8_config3_shard_rows():
    """BASELINE configs[2] as the driver launches it -- 8 ranks, 1,024 frames per rank -- without GPUs: every rank fabricates its
    shard's lists, the gathered tensor must hold 8 x 1,024 rows with rank r's frames at rows [1024 r, 1024 (r + 1)) in frame
    order and the true counts (also those above the gather capacity) in the count word."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--cpu-dry-run", "--steps", "1", "--warmup", "0",
                        "--frames", "1024"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["dry_run"] is True and d["ranks_seen"] == list(range(8))
    assert d["gathered_rows"] == 8192 and d["gather_ok"] is True


def test_comm_abort_on_a_plain_handle_and_header_declares_it():
    """pigo_comm_abort exists in the C ABI (include/pigo_hip.h) and is a no-op on a world-1 handle without RCCL."""
    c = distributed.Comm(0, 1, 0)
    c.abort()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert "pigo_status pigo_comm_abort(pigo_comm *c);" in open(os.path.join(root, "include", "pigo_hip.h")).read()


def test_pack_unpack_roundtrip():
    dets, counts = _fake_lists(0, 5, 12)
    wire = distributed.pack_lists(dets, counts, 8)
    lists, cnt = distributed.unpack_lists(wire, 8)
    assert cnt.tolist() == counts.tolist()  # the true counts survive, even when > gather_cap
    # the flags word: cut at the gather capacity (count > 8), cut at the lists' capacity (count > 12) -- torch and C ABI agree
    fl = distributed.row_flags(wire)
    for f in range(5):
        want = (distributed.WIRE_TRUNCATED_GATHER if int(counts[f]) > 8 else 0) | (distributed.WIRE_TRUNCATED_DETCAP if int(counts[f]) > 12 else 0)
        assert int(fl[f]) == want, (f, int(counts[f]), int(fl[f]))
    host = distributed.pack_lists_host(dets.numpy().view(core.DET_DTYPE).reshape(5, 12), counts.numpy(), 6, 8)
    assert (host[:5] == wire.numpy()).all() and int(host[5, 1]) == distributed.WIRE_PADDING
    # a rank whose scan overflowed a queue or met a would-panic frame says so in every row
    fl2 = distributed.row_flags(distributed.pack_lists(dets, counts, 8, plan_flags=(2, 1)))
    assert all(int(v) & distributed.WIRE_QUEUE_OVERFLOW and int(v) & distributed.WIRE_WOULD_PANIC for v in fl2)
    # clusters made from a raw list that was cut at det_cap are incomplete as well
    fl3 = distributed.row_flags(distributed.pack_lists(dets, counts.clamp(max=3), 8, raw_counts=counts))
    assert [bool(int(v) & distributed.WIRE_TRUNCATED_DETCAP) for v in fl3] == [int(c) > 12 for c in counts]
    for f, l in enumerate(lists):
        assert len(l) == min(int(counts[f]), 8) and l.dtype == core.DET_DTYPE
    import pytest
    with pytest.raises(ValueError):
        distributed.pack_lists(dets, counts, 16)  # gather_cap larger than the lists' capacity


def test_wire_flag_constants_match_the_header():
    """The Python mirror's WIRE_* bits are the header's PIGO_WIRE_* (include/pigo_hip.h is what a cgo / C++ host binds), and a row
    is two head words + gather_cap records in both."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "pigo_hip.h")).read()
    want = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define PIGO_WIRE_(\w+) (\d+)", hdr)}
    assert want == {"TRUNCATED_GATHER": distributed.WIRE_TRUNCATED_GATHER, "TRUNCATED_DETCAP": distributed.WIRE_TRUNCATED_DETCAP,
                    "QUEUE_OVERFLOW": distributed.WIRE_QUEUE_OVERFLOW, "WOULD_PANIC": distributed.WIRE_WOULD_PANIC,
                    "RANK_FAILED": distributed.WIRE_RANK_FAILED, "PADDING": distributed.WIRE_PADDING}, want
    L = core.load_library()
    for g in (1, 8, 64):
        assert int(L.pigo_wire_words(g)) == distributed.WIRE_HEAD + 4 * g
    assert "pigo_wire_row_flags" in open(os.path.join(root, "include", "pigo.hpp")).read()
