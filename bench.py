#!/usr/bin/env python3
"""bench.py -- the 1080p facefinder scan benchmark (BASELINE.json metric: Mwindows/s and frames/s; % HBM roofline).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with no launcher around it starts the N ranks itself (re-exec under torch.distributed.run); under a launcher
WORLD_SIZE must equal N.

One STEP = one pass of the hot path over one batch of synthetic 1080p frames that are already resident in
HBM: RunCascade (scan + order restore) for every frame, per-frame ClusterDetections on the GPU, and -- for
N > 1 -- one RCCL all-gather of the fixed-capacity per-frame cluster lists, issued by libpigo_hip.so itself
(pigo_run_batch_sharded, include/pigo_hip.h).  After the timed region the first frames of the timed batch are checked
against the CPU oracle (`verified_frames`) and the run fails if any frame overflowed det_cap.  Workload = BASELINE.json
configs[1] (1920x1080, facefinder, MinSize 20, MaxSize 1000, ShiftFactor 0.1, ScaleFactor 1.1) applied to a
batch of `--frames` seeded SYN-FACES frames per GPU (weak scaling: per-GPU work is fixed as N grows).

Rank 0 prints ONE JSON line (see the task contract) with two extra objects:
  roofline      HBM roofline of the scan (k_scan_region's two launches with the big scales' side chain k_scan_big ->
                k_big_pool -> k_tail_deep running next to them): algorithmic bytes per step (every frame read once +
                16 B per detection) / the timed step; the per-kernel HIP-event times (each launch alone on the launch
                stream) are in kernel_ms; peak 8.0 TB/s.
  single_frame  BASELINE configs[1] taken literally: one 1080p frame per call, HBM-resident and from a host buffer.
  puploc        side measurement of the RunDetector kernel (4096 requests x 63 perturbations): requests/s.
  gray          side measurement of the RgbToGrayscale kernel (the streaming step in front of the scan): GB/s vs 8 TB/s.
  cpu_baseline  the CPU oracle (a C restatement of the reference's Go path -- the Go toolchain is absent) timed
                on this host's cores on a bounded sample of the same frames.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=128, help="frames per GPU per step (resident in HBM)")
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--min-size", type=int, default=20)
    ap.add_argument("--max-size", type=int, default=1000)
    ap.add_argument("--shift", type=float, default=0.1)
    ap.add_argument("--scale", type=float, default=1.1)
    ap.add_argument("--angle", type=float, default=0.0)
    ap.add_argument("--iou", type=float, default=0.2)
    ap.add_argument("--kind", choices=["faces", "noise"], default="faces")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--det-cap", type=int, default=1024)
    ap.add_argument("--gather-cap", type=int, default=64)
    ap.add_argument("--variant", type=int, default=None, help="0 = monolithic, 2 = LDS tiles, 3 = LDS regions + big-scale side chain / k_scan_one (default: the plan's choice, 3)")
    ap.add_argument("--no-cluster", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even for one rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gray", action="store_true", help="skip the side measurements (RgbToGrayscale, RunDetector, single frame)")
    ap.add_argument("--no-kernel-times", action="store_true", help="skip the per-kernel HIP-event pass (a rocprofv3 trace of the run then only holds overlapped steps)")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the one-frame-per-call leg (keeps kernel profiles per-batch)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU sample (0 = auto, ~15 s)")
    ap.add_argument("--face-rotation", type=float, default=0.0, help="rotate the pasted face patches by this many degrees (-79: what a scan at --angle 0.8 detects)")
    ap.add_argument("--verify-frames", type=int, default=128,
                    help="frames of the timed batch checked against the CPU oracle afterwards (default: every frame of the default batch; "
                         "one host thread per frame, ~0.25 s of one core each)")
    ap.add_argument("--gather", choices=["cabi", "torch"], default="cabi",
                    help="N > 1: all-gather through the C ABI (pigo_run_batch_sharded -> ncclAllGather) or torch.distributed")
    ap.add_argument("--shard-frames", type=int, default=1024,
                    help="frames of the config-3 shard leg (BASELINE configs[2]: 1024 frames per GPU); 0 = skip")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the config-4 (rotated), config-5 (4K) and reference-benchmark legs")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="no GPU: fabricated lists + gloo all-gather; checks the launch / sharding / gather plumbing of --gpus N")
    return ap.parse_args()


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run and hand
    their single JSON line through.  Under a launcher the world size must match --gpus."""
    import socket
    import subprocess
    world = os.environ.get("WORLD_SIZE")
    if world is not None:
        if int(world) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        return
    if args.gpus <= 1:
        return
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def cpu_dry_run(args, json_fd):
    """The N-rank plumbing without a GPU: every rank fabricates its shard's lists, packs them into the wire format and
    all-gathers them with gloo; rank 0 prints the record.  Not a measurement."""
    import torch
    import torch.distributed as dist
    from pigo_amd import core, distributed
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, gcap = args.frames, min(args.gather_cap, args.det_cap)
    counts = np.array([(rank * B + f) % (gcap + 2) for f in range(B)], dtype=np.int32)
    dets = np.zeros((B, args.det_cap), dtype=core.DET_DTYPE)
    for f in range(B):
        dets[f, :min(counts[f], args.det_cap)]["row"] = rank * B + f
    out = None
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        wire = torch.from_numpy(distributed.pack_lists_host(dets, counts, B, gcap))
        out = torch.empty((world * B, wire.shape[1]), dtype=torch.int32)
        dist.all_gather_into_tensor(out, wire)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    seen = [None] * world
    dist.all_gather_object(seen, rank)
    ok = all(int(out[r * B + f, 0]) == (r * B + f) % (gcap + 2) for r in range(world) for f in range(B))
    # ... and the flags word of every row: a list longer than the gather capacity is marked truncated, nothing else is raised
    ok = ok and all(int(out[r * B + f, 1]) == (distributed.WIRE_TRUNCATED_GATHER if (r * B + f) % (gcap + 2) > gcap else 0)
                    for r in range(world) for f in range(B))
    if rank == 0:
        os.write(json_fd, (json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "unit": "Mwindows/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / (args.warmup + args.steps) * 1e3, 4), "dry_run": True,
                          "ranks_seen": sorted(seen), "gathered_rows": int(out.shape[0]), "gather_ok": bool(ok)}) + "\n").encode())
    dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def cpu_baseline(args, frames, windows_per_frame):
    """Time the oracle (C restatement of core/pigo.go:113-258, -O2) on this host: one thread (the reference's
    RunCascade is a single goroutine) and all cores (one frame per thread)."""
    import threading
    import oracle
    from pigo_amd import synth
    orc = oracle.OraclePigo.unpack(synth.facefinder_bytes())
    ncores = os.cpu_count() or 1

    def scan(f):
        return orc.run_cascade(f, args.rows, args.cols, args.cols, args.min_size, args.max_size, args.shift, args.scale, args.angle)

    t = time.perf_counter()
    d0 = scan(frames[0])
    one = time.perf_counter() - t  # also the warm-up
    # ClusterDetections on that frame's list, timed separately (mirrors BenchmarkPigoClusterDetection, core/pigo_test.go:115-143)
    creps = 20
    t = time.perf_counter()
    for _ in range(creps):
        orc.cluster_detections(d0.copy(), args.iou)
    cluster_ms = (time.perf_counter() - t) / creps * 1e3
    n1 = max(1, min(len(frames), int(5.0 / max(one, 1e-3))))
    t = time.perf_counter()
    for i in range(n1):
        scan(frames[i % len(frames)])
    t1 = (time.perf_counter() - t) / n1
    # all cores: ctypes releases the GIL during the C call.  Calibrate with one frame per thread, then size the timed
    # sample to ~10 s so that the whole leg stays within ~25 s of CPU wall time.
    def run_all(per_thread):
        threads = [threading.Thread(target=lambda k=k: [scan(frames[(k + j) % len(frames)]) for j in range(per_thread)]) for k in range(ncores)]
        t = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        return time.perf_counter() - t

    tcal = run_all(1)
    per_thread = args.cpu_frames or max(1, min(8, int(10.0 / max(tcal, 1e-3))))
    tall = run_all(per_thread)
    fps_all = ncores * per_thread / tall
    return {
        "value": round(fps_all * windows_per_frame / 1e6, 3), "unit": "Mwindows/s", "cores": ncores, "kind": "port",
        "sample": f"{ncores} threads x {per_thread} of the benchmark's {args.rows}x{args.cols} frames, one frame per thread "
                  f"(C oracle, gcc -O2; the Go reference cannot run here)",
        "frames_per_s": round(fps_all, 3),
        "single_thread": {"value": round(windows_per_frame / t1 / 1e6, 3), "unit": "Mwindows/s", "ms_per_frame": round(t1 * 1e3, 2),
                          "frames": n1, "cluster_ms_per_frame": round(cluster_ms, 4), "detections_in_that_frame": int(len(d0))},
    }


def verify_frame_indices(nframes, k):
    """k frames of a batch to check: the first ceil(k/2) and the last floor(k/2) -- the head of the XCD dealing and its tail
    (the `nframes % 8` remainder blocks of map_block, the last pipeline chunk)."""
    k = min(k, nframes)
    head = (k + 1) // 2
    return sorted(set(range(head)) | set(range(nframes - (k - head), nframes)))


def verify_against_oracle(args, frames, dets, counts, clusters, ccounts, k, what="batch"):
    """Bit-exact check of k frames of the timed batch -- its first and its last ones (verify_frame_indices) -- raw RunCascade lists
    and clusters, against the CPU oracle, one host thread per frame.  Raises on any difference: a fast wrong answer must not
    produce a bench line.  Returns the frame indices checked."""
    import threading
    import oracle
    from pigo_amd import batch, synth
    orc = oracle.OraclePigo.unpack(synth.facefinder_bytes())
    idx = verify_frame_indices(len(frames), k)
    sel = torch_index(idx, dets.device)
    got = batch.dets_to_numpy(dets[sel], counts[sel])
    gcl = batch.dets_to_numpy(clusters[sel], ccounts[sel]) if clusters is not None else None
    k = len(idx)
    want, wantc = [None] * k, [None] * k

    def work(j):
        f = idx[j]
        want[j] = orc.run_cascade(frames[f], args.rows, args.cols, args.cols, args.min_size, args.max_size, args.shift, args.scale, args.angle)
        wantc[j] = orc.cluster_detections(want[j].copy(), args.iou)

    th = [threading.Thread(target=work, args=(j,)) for j in range(k)]
    for t in th:
        t.start()
    for t in th:
        t.join()

    def same(a, b, what):
        if len(a) != len(b):
            raise SystemExit(f"bench.py: VERIFICATION FAILED: {what}: {len(a)} records, oracle {len(b)}")
        for i in range(len(a)):
            if (int(a[i]["row"]), int(a[i]["col"]), int(a[i]["scale"])) != (int(b[i]["row"]), int(b[i]["col"]), int(b[i]["scale"])) or \
                    np.float32(a[i]["q"]) != np.float32(b[i]["q"]):
                raise SystemExit(f"bench.py: VERIFICATION FAILED: {what} record {i}: {a[i]} vs oracle {b[i]}")

    for j in range(k):
        same(got[j], want[j], f"{what} frame {idx[j]} RunCascade")
        if gcl is not None:
            same(gcl[j], wantc[j], f"{what} frame {idx[j]} ClusterDetections")
    return idx


def config_leg(args, pg, dev, what, frames_n, steps, verify_k, **over):
    """One more BASELINE configuration as a side leg of the default line: the same step (RunCascade + ClusterDetections on
    HBM-resident frames) with other plan parameters, timed over `steps` steps after three warm-up steps, `verify_k` frames of the
    batch checked bit-exactly against the CPU oracle."""
    import argparse
    import torch
    from pigo_amd import batch, synth
    a2 = argparse.Namespace(**{**vars(args), **over})
    fr = synth.make_frames(a2.kind, frames_n, a2.rows, a2.cols, seed=a2.seed, first_index=0, rotate_deg=a2.face_rotation)
    d_fr = torch.from_numpy(fr).to(dev)
    plan = batch.ScanPlan(pg, a2.rows, a2.cols, MinSize=a2.min_size, MaxSize=a2.max_size, ShiftFactor=a2.shift, ScaleFactor=a2.scale,
                          angle=a2.angle, max_frames=frames_n, det_cap=a2.det_cap)
    info = plan.info()
    dets, counts = plan.alloc_outputs(frames_n)
    cl = plan.alloc_cluster_outputs(dets, counts)

    def step():
        plan.run(d_fr, dets, counts)
        plan.cluster(dets, counts, a2.iou, out=cl)

    for _ in range(3):  # (the GPU has idled through the previous leg's CPU verification: three steps bring its clocks back)
        step()
    torch.cuda.synchronize()
    plan.status()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    plan.status()
    if int(counts.max().item()) > a2.det_cap:
        raise SystemExit(f"bench.py: {what}: a frame has {int(counts.max().item())} detections, det_cap is {a2.det_cap}")
    checked = verify_against_oracle(a2, fr, dets, counts, cl[1], cl[2], verify_k, what=what) if verify_k > 0 else []
    wpf = int(info.windows_per_frame)
    leg = {"workload": f"{a2.cols}x{a2.rows} SYN-{a2.kind.upper()}" + (f", faces rotated by {a2.face_rotation} deg" if a2.face_rotation else "") +
                       f", MinSize={a2.min_size} MaxSize={a2.max_size} Shift={a2.shift} Scale={a2.scale} angle={a2.angle}, {frames_n} HBM-resident frames per step",
           "frames": frames_n, "steps": steps, "ms_per_step": round(ms, 4), "windows_per_frame": wpf, "variant": int(info.variant),
           "mwindows_per_s": round(frames_n * wpf / ms / 1e3, 1), "frames_per_s": round(frames_n / ms * 1e3, 1),
           "detections": int(counts.sum().item()), "clusters": int(cl[2].sum().item()), "verified_frames": checked,
           "verified": ("every frame of the leg's batch" if len(checked) == frames_n else f"{len(checked)} of {frames_n} frames") + " vs the CPU oracle, raw lists and clusters bit-exact",
           # the same definition as the headline's roofline: every frame read once + 16 B per emitted detection, over the timed step
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "algorithmic_bytes_per_window": round(a2.rows * a2.cols / wpf, 4),
                        "achieved": round((frames_n * a2.rows * a2.cols + 16 * int(counts.sum().item())) / (ms * 1e-3) / 1e9, 2),
                        "frac": round((frames_n * a2.rows * a2.cols + 16 * int(counts.sum().item())) / (ms * 1e-3) / 1e9 / 8000.0, 5), "traffic": None}}
    del plan, d_fr, dets, counts, cl
    return leg


def reference_benchmark_leg(args, pg):
    """The reference's own in-tree benchmark, BenchmarkPigoFaceDetection (core/pigo_test.go:96-118): RunCascade + ClusterDetections
    (IoU 0.1) on testdata/sample.jpg at MinSize 20, MaxSize 1000, ShiftFactor 0.2, ScaleFactor 1.1 -- here on the committed gray
    fixture of that image (Go's JPEG decoder is not reproducible without Go), one call per iteration through the drop-in
    single-frame API (host buffer in, host lists out), beside the CPU oracle on the same input."""
    import oracle
    from pigo_amd import core, synth
    gray = synth.sample_gray()
    rows, cols = gray.shape
    cp = core.CascadeParams(MinSize=20, MaxSize=1000, ShiftFactor=0.2, ScaleFactor=1.1,
                            ImageParams=core.ImageParams(Pixels=gray, Rows=rows, Cols=cols, Dim=cols))
    for _ in range(5):
        d = pg.RunCascade(cp, 0.0)
        c = pg.ClusterDetections(d, 0.1)
    n = 200
    t = time.perf_counter()
    for _ in range(n):
        d = pg.RunCascade(cp, 0.0)
        c = pg.ClusterDetections(d, 0.1)
    gpu_ms = (time.perf_counter() - t) / n * 1e3
    t = time.perf_counter()
    for _ in range(n):
        d = pg.RunCascade(cp, 0.0)
    scan_ms = (time.perf_counter() - t) / n * 1e3
    orc = oracle.OraclePigo.unpack(synth.facefinder_bytes())
    want = orc.run_cascade(gray, rows, cols, cols, 20, 1000, 0.2, 1.1, 0.0)
    wantc = orc.cluster_detections(want.copy(), 0.1)
    ok = len(d) == len(want) and all((int(a["row"]), int(a["col"]), int(a["scale"]), np.float32(a["q"])) ==
                                     (int(b["row"]), int(b["col"]), int(b["scale"]), np.float32(b["q"])) for a, b in zip(d, want))
    ok = ok and len(c) == len(wantc) and all((int(a["row"]), int(a["col"]), int(a["scale"]), np.float32(a["q"])) ==
                                             (int(b["row"]), int(b["col"]), int(b["scale"]), np.float32(b["q"])) for a, b in zip(c, wantc))
    if not ok:
        raise SystemExit("bench.py: VERIFICATION FAILED: reference benchmark leg (sample fixture) differs from the oracle")
    m = 20
    t = time.perf_counter()
    for _ in range(m):
        w = orc.run_cascade(gray, rows, cols, cols, 20, 1000, 0.2, 1.1, 0.0)
        orc.cluster_detections(w.copy(), 0.1)
    cpu_ms = (time.perf_counter() - t) / m * 1e3
    return {"benchmark": "BenchmarkPigoFaceDetection (core/pigo_test.go:96-118): RunCascade(20/1000/0.2/1.1) + ClusterDetections(0.1), sample fixture "
                         f"{cols}x{rows}, one call per iteration, host buffers",
            "gpu_ms_per_op": round(gpu_ms, 4), "gpu_scan_only_ms_per_op": round(scan_ms, 4), "cpu_oracle_ms_per_op": round(cpu_ms, 3),
            "cpu_kind": "port (C oracle, one thread; a `go test -bench` figure can be put beside gpu_ms_per_op)",
            "detections": int(len(d)), "clusters": int(len(c)), "verified": True}


def torch_index(idx, device):
    import torch
    return torch.tensor(idx, dtype=torch.long, device=device)


def main():
    args = parse_args()
    spawn_ranks_if_needed(args)
    # stdout carries exactly ONE line, the JSON record: native libraries (RCCL and gloo print banners to stdout) and
    # anything else that writes to fd 1 go to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.cpu_dry_run:
        return cpu_dry_run(args, json_fd)
    import torch
    import torch.distributed as dist
    from pigo_amd import batch, core, distributed, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    n_gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B = args.frames
    frames = synth.make_frames(args.kind, B, args.rows, args.cols, seed=args.seed, first_index=rank * B, rotate_deg=args.face_rotation)
    d_frames = torch.from_numpy(frames).to(dev)

    pg = core.NewPigo(local_rank).Unpack(synth.facefinder_bytes())
    plan = batch.ScanPlan(pg, args.rows, args.cols, MinSize=args.min_size, MaxSize=args.max_size, ShiftFactor=args.shift,
                          ScaleFactor=args.scale, angle=args.angle, max_frames=B, det_cap=args.det_cap)
    if args.variant is not None:
        plan.set_variant(args.variant)
    info = plan.info()
    dets, counts = plan.alloc_outputs(B)
    gcap = min(args.gather_cap, args.det_cap)
    cl_out = plan.alloc_cluster_outputs(dets, counts)

    # N > 1: the all-gather belongs to the step.  Default: the C ABI's own RCCL communicator (what a Go / C++ host would
    # use); torch.distributed only ships the 128-byte RCCL id.  If that communicator cannot be created the run says so
    # on stderr and in the record and uses torch.distributed's all-gather (RCCL as well) instead.
    comm, gather_mode, gathered = None, None, None
    if use_dist:
        gather_mode = args.gather
        if gather_mode == "cabi":
            try:
                comm = distributed.Comm.from_torch(local_rank, force_rccl=True)  # world 1 too: a real one-rank RCCL communicator
            except Exception as e:  # noqa: BLE001 -- reported, not swallowed
                print(f"[rank {rank}] pigo_comm_init failed ({e}); using torch.distributed for the all-gather", file=sys.stderr)
                comm = None
            flag = torch.tensor([1 if comm is not None else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                comm, gather_mode = None, "torch (pigo_comm_init failed)"
            else:  # which collective the C ABI will issue: ncclAllGather, or (world 1 without an id) a device copy
                gather_mode = "cabi/rccl" if comm.uses_rccl else "cabi/memcpy"
        gathered = torch.zeros((world * B, 2 + 4 * gcap), dtype=torch.int32, device=dev)

    def step():
        if comm is not None:  # scan + cluster + pack + ncclAllGather, all enqueued by libpigo_hip.so
            return distributed.run_batch_sharded(plan, comm, d_frames, B, -1.0 if args.no_cluster else args.iou, gcap, out=gathered)
        plan.run(d_frames, dets, counts)
        if args.no_cluster:
            lists, lcounts = dets, counts
        else:
            _, lists, lcounts, _ = plan.cluster(dets, counts, args.iou, out=cl_out)
        if use_dist:
            return distributed.allgather_lists(lists, lcounts, gcap, B)
        return lists

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    plan.status()  # a queue overflow or a would-panic frame invalidates the run: fail loudly
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    last = step() if use_dist else None  # (untimed) one more gathered result for the checks below
    torch.cuda.synchronize()
    plan.status()   # queue overflow, would-panic frame or a frame with more than det_cap detections: no bench line
    if comm is not None:  # the sharded entry point keeps its lists inside the plan: run the plain path once for the checks
        plan.run(d_frames, dets, counts)
        if not args.no_cluster:
            plan.cluster(dets, counts, args.iou, out=cl_out)
        torch.cuda.synchronize()
        plan.status()
    if int(counts.max().item()) > args.det_cap:
        raise SystemExit(f"bench.py: a frame has {int(counts.max().item())} detections, det_cap is {args.det_cap}: truncated lists")
    verified, verified_idx = 0, []
    if rank == 0 and args.verify_frames > 0:
        verified_idx = verify_against_oracle(args, frames, dets, counts, None if args.no_cluster else cl_out[1],
                                             None if args.no_cluster else cl_out[2], min(args.verify_frames, B))
        verified = len(verified_idx)
        if last is not None:  # rank 0's own rows of the gathered tensor are its cluster lists in wire format
            ref = distributed.pack_lists(dets if args.no_cluster else cl_out[1], counts if args.no_cluster else cl_out[2], gcap)
            if not torch.equal(last[:B], ref):
                raise SystemExit("bench.py: VERIFICATION FAILED: gathered rows of rank 0 differ from its lists")
            if world > 1:  # ... and a frame scanned by ANOTHER rank, as it arrived through the all-gather, against the oracle
                import oracle
                fo = synth.make_frames(args.kind, 1, args.rows, args.cols, seed=args.seed, first_index=(world - 1) * B, rotate_deg=args.face_rotation)[0]
                orc = oracle.OraclePigo.unpack(synth.facefinder_bytes())
                w = orc.run_cascade(fo, args.rows, args.cols, args.cols, args.min_size, args.max_size, args.shift, args.scale, args.angle)
                if not args.no_cluster:
                    w = orc.cluster_detections(w.copy(), args.iou)
                g, cnt = distributed.unpack_list_host(last[(world - 1) * B].cpu().numpy(), gcap)
                if cnt != len(w) or any((int(a["row"]), int(a["col"]), int(a["scale"]), np.float32(a["q"])) !=
                                        (int(b["row"]), int(b["col"]), int(b["scale"]), np.float32(b["q"])) for a, b in zip(g, w[:gcap])):
                    raise SystemExit(f"bench.py: VERIFICATION FAILED: frame {(world - 1) * B} (rank {world - 1}) through the all-gather")
                verified += 1
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- per-kernel timing (HIP events on the launch stream), outside the timed region ----
    plan.set_profiling(True)
    ktimes = {}
    reps = 5
    for _ in range(0 if args.no_kernel_times else reps):
        plan.run(d_frames, dets, counts)
        torch.cuda.synchronize()
        for name, ms in plan.last_timings():
            ktimes[name] = ktimes.get(name, 0.0) + ms / reps
    plan.set_profiling(False)
    survivors = plan.last_queue_count()
    if os.environ.get("PIGO_DEBUG_STATS"):
        st = plan.debug_stats()
        tiles = max(st[4], 1)
        print("debug_stats (cycles per tile): copy %.0f stage0 %.0f dense %.0f late %.0f | late windows/tile %.1f late trees/tile %.1f tiles %d" %
              (st[0] / tiles, st[1] / tiles, st[2] / tiles, st[3] / tiles, st[5] / tiles, st[6] / tiles, st[4]), file=sys.stderr)
        print("debug_stats raw:", st, file=sys.stderr)
        ne = max(st[8], 1)
        print("debug_stats tail_deep (wave 0): %.0f cycles per entry | passes/entry %.2f entries %d" % (st[9] / ne, st[14] / ne, st[8]), file=sys.stderr)
    ndet = int(counts.sum().item())
    cev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    cluster_ms = None
    if not args.no_cluster:
        cev[0].record()
        for _ in range(reps):
            plan.cluster(dets, counts, args.iou, out=cl_out)
        cev[1].record()
        torch.cuda.synchronize()
        cluster_ms = cev[0].elapsed_time(cev[1]) / reps

    # ---- BASELINE configs[1] as stated: ONE 1080p frame.  (a) resident in HBM, back-to-back launches of a one-frame plan (ONE
    # kernel launch per call: k_scan_one) -- first NEXT TO the live batch plan of the timed workload (its workspace and streams still
    # there: until round 4 a one-frame call forked onto a side stream of its own and ran 0.22-0.26 ms in that company against 0.145
    # alone), then alone; (b) RunCascade on a host buffer: H2D of the frame, scan, results written to pinned host memory by the scan
    # itself (PCIe-inclusive; never `value`)
    single_leg = None
    side_legs = rank == 0 and n_gpus == 1 and not args.no_gray
    one_next_ms = None
    if side_legs and not args.no_single_frame:
        plan1 = batch.ScanPlan(pg, args.rows, args.cols, MinSize=args.min_size, MaxSize=args.max_size, ShiftFactor=args.shift,
                               ScaleFactor=args.scale, angle=args.angle, max_frames=1, det_cap=args.det_cap)
        d1, c1 = plan1.alloc_outputs(1)

        def time_one(n=100):
            for _ in range(5):
                plan1.run(d_frames[:1], d1, c1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                plan1.run(d_frames[:1], d1, c1)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / n * 1e3
            plan1.status()
            return ms

        one_next_ms = time_one()
        # (the one-frame plan's list for frame 0 is the batch plan's, record for record)
        if int(c1[0].item()) != int(counts[0].item()) or not torch.equal(d1[0, :int(c1[0].item())], dets[0, :int(counts[0].item())]):
            raise SystemExit("bench.py: VERIFICATION FAILED: the one-frame plan's detections for frame 0 differ from the batch plan's")
    # The timed workload's plan (workspace, streams) goes before the other side legs build theirs: every leg as in a process of its own.
    del plan
    if side_legs and not args.no_single_frame:
        dev_ms = time_one()
        cp1 = core.CascadeParams(MinSize=args.min_size, MaxSize=args.max_size, ShiftFactor=args.shift, ScaleFactor=args.scale,
                                 ImageParams=core.ImageParams(Pixels=frames[0], Rows=args.rows, Cols=args.cols, Dim=args.cols))
        for _ in range(3):
            pg.RunCascade(cp1, args.angle)
        t1 = time.perf_counter()
        for _ in range(20):
            pg.RunCascade(cp1, args.angle)
        host_ms = (time.perf_counter() - t1) / 20 * 1e3
        w1 = int(info.windows_per_frame)
        single_leg = {"hbm_resident_ms": round(dev_ms, 4), "hbm_resident_mwindows_per_s": round(w1 / dev_ms / 1e3, 1),
                      "hbm_resident_next_to_live_batch_plan_ms": round(one_next_ms, 4),
                      "host_buffer_ms": round(host_ms, 4), "host_buffer_mwindows_per_s": round(w1 / host_ms / 1e3, 1),
                      "launches_per_call": 1,
                      "note": "one frame per call, ONE kernel launch (k_scan_one); frame 0's list checked against the batch plan's; "
                              "host_buffer includes PCIe H2D, the results written to pinned host memory by the launch, and one synchronisation"}
        del plan1

    # ---- side measurement, outside the timed region: RgbToGrayscale (core/grayscale.go:8-23), the streaming step in
    # front of the scan.  RGBA frames {g,g,g,255} built on the GPU from the gray batch; the kernel must give them back.
    gray_leg = None
    if side_legs:
        gn = min(B, 64)
        rgba = torch.empty((gn, args.rows, args.cols, 4), dtype=torch.uint8, device=dev)
        rgba[..., :3] = d_frames[:gn].unsqueeze(-1)
        rgba[..., 3] = 255
        gout = torch.zeros((gn, args.rows, args.cols), dtype=torch.uint8, device=dev)
        for _ in range(3):
            batch.rgb_to_grayscale(rgba, kind=core.PIX_NRGBA, out=gout)
        gev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        greps = 20
        gev[0].record()
        for _ in range(greps):
            batch.rgb_to_grayscale(rgba, kind=core.PIX_NRGBA, out=gout)  # launched on torch's current stream
        gev[1].record()
        torch.cuda.synchronize()
        gms = gev[0].elapsed_time(gev[1]) / greps
        gbytes = gn * args.rows * args.cols * 5  # 4 B read + 1 B written per pixel
        gray_leg = {"kernel": "k_rgb_to_gray_lin<NRGBA>", "frames": gn, "ms_per_launch": round(gms, 4), "bytes_per_launch": gbytes,
                    "bound": "hbm", "achieved": round(gbytes / (gms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbytes / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "roundtrip_ok": bool(torch.equal(gout, d_frames[:gn]))}
        del rgba, gout

    # ---- side measurement: RunDetector (core/puploc.go:239-277), the step behind ClusterDetections.  4096 eye-sized
    # requests with 63 perturbations each, spread over the resident frames; one launch, one workgroup per request.
    pup_leg = None
    if side_legs:
        plc = core.NewPuplocCascade(local_rank).UnpackCascade(synth.cascade_bytes("puploc"))
        nreq = 4096
        rng = np.random.default_rng(args.seed)
        reqs = np.zeros(nreq, dtype=core.PUPLOC_REQ_DTYPE)
        reqs["row"], reqs["col"] = rng.integers(40, args.rows - 40, nreq), rng.integers(40, args.cols - 40, nreq)
        reqs["scale"] = rng.uniform(10, 60, nreq).astype(np.float32)
        reqs["perturbs"] = 63
        reqs["frame"] = rng.integers(0, B, nreq)
        rnd = rng.random((nreq, 189), dtype=np.float32)
        d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(nreq, 24)).to(dev)
        d_rnd = torch.from_numpy(rnd).to(dev)
        pout = torch.zeros((nreq, 4), dtype=torch.int32, device=dev)
        for _ in range(2):
            batch.puploc_run_batch(plc, d_frames, d_reqs, d_rnd, out=pout)
        pev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        preps = 10
        pev[0].record()
        for _ in range(preps):
            batch.puploc_run_batch(plc, d_frames, d_reqs, d_rnd, out=pout)
        pev[1].record()
        torch.cuda.synchronize()
        batch.puploc_status(plc)
        pms = pev[0].elapsed_time(pev[1]) / preps
        # bound: every level of every tree walk is one dependent round trip to L2 (codes + two pixels); nothing streams.
        # Algorithmic bytes per request: 63 perturbations x 100 trees x 10 levels x (4 B code + 2 B pixels) + 100 x 63 x 8 B
        # leaves = 428 KB of cache-line-granular gathers out of a 1.2 MB table, i.e. latency, not bandwidth
        pbytes = nreq * (63 * 100 * 10 * 6 + 63 * 100 * 8)
        pup_leg = {"kernel": "k_puploc", "requests": nreq, "perturbs": 63, "ms_per_launch": round(pms, 4),
                   "requests_per_s": round(nreq / (pms * 1e-3), 1), "tree_walks_per_s": round(nreq * 63 * 100 / (pms * 1e-3), 1),
                   "bound": "latency (dependent L2 gathers: 10 levels x 100 trees per perturbation, one round trip per level)",
                   "gather_bytes_per_launch": pbytes, "gather_gbs": round(pbytes / (pms * 1e-3) / 1e9, 1),
                   "dependent_round_trips_per_request": 5 * 10, "note": "5 stages x 10 levels are serial per perturbation; "
                   "63 perturbations x 20 trees of a stage run in parallel lanes"}
        if not args.no_cpu_baseline:
            import oracle
            oplc = oracle.OraclePuploc.unpack(synth.cascade_bytes("puploc"))
            t = time.perf_counter()
            for i in range(40):
                r = reqs[i]
                oplc.run_detector(int(r["row"]), int(r["col"]), float(r["scale"]), 63, frames[r["frame"]], args.rows, args.cols, args.cols, 0.0,
                                  False, rnd[i], None)
            pup_leg["cpu_requests_per_s_one_thread"] = round(40 / (time.perf_counter() - t), 1)

    # ---- BASELINE configs[2]'s per-GPU shard: 1024 resident frames (2.1 GB) per step; same plan parameters, seeded frames
    shard_leg = None
    if side_legs and args.shard_frames > B and (args.rows, args.cols) == (1080, 1920):
        S = args.shard_frames
        fS = synth.make_frames(args.kind, S, args.rows, args.cols, seed=args.seed, first_index=0, rotate_deg=args.face_rotation)
        dS = torch.from_numpy(fS).to(dev)
        planS = batch.ScanPlan(pg, args.rows, args.cols, MinSize=args.min_size, MaxSize=args.max_size, ShiftFactor=args.shift,
                               ScaleFactor=args.scale, angle=args.angle, max_frames=S, det_cap=args.det_cap)
        detS, cntS = planS.alloc_outputs(S)
        clS = planS.alloc_cluster_outputs(detS, cntS)

        def stepS():
            planS.run(dS, detS, cntS)
            planS.cluster(detS, cntS, args.iou, out=clS)

        for _ in range(3):  # (the GPU has idled through the CPU-side generation of 1,024 frames: three steps bring its clocks back)
            stepS()
        torch.cuda.synchronize()
        tS = time.perf_counter()
        for _ in range(5):
            stepS()
        torch.cuda.synchronize()
        msS = (time.perf_counter() - tS) / 5 * 1e3
        planS.status()
        assert int(cntS.max().item()) <= args.det_cap
        assert torch.equal(cntS[:B], counts) and torch.equal(detS[:B], dets), "the shard's first frames must reproduce the default batch"
        shard_checked = []
        if args.verify_frames > 0:  # frames only the shard has (its last ones), against the CPU oracle
            tail = S - 16
            shard_checked = [tail + j for j in verify_against_oracle(args, fS[tail:], detS[tail:], cntS[tail:], clS[1][tail:], clS[2][tail:], 16, what="config-3 shard tail")]
        shard_leg = {"frames_per_gpu": S, "ms_per_step": round(msS, 3), "mwindows_per_s": round(S * int(info.windows_per_frame) / msS / 1e3, 1),
                     "frames_per_s": round(S / msS * 1e3, 1), "resident_bytes": int(fS.nbytes), "detections": int(cntS.sum().item()),
                     "verified_frames": shard_checked,
                     "note": "BASELINE configs[2] per-GPU shard (8192 frames / 8 GPUs); its first frames compared with the default batch, its last sixteen with the CPU oracle"}
        del planS, dS, detS, cntS, clS, fS

    # ---- BASELINE configs[3] (rotated scan, angle 0.8: on the benchmark's upright faces and on faces rotated the way that scan
    # finds them) and configs[4] (4K stress ladder) as side legs, and the reference's own in-tree benchmark -- driver-visible
    config4_leg = config5_leg = ref_leg = None
    default_cfg = (args.rows, args.cols, args.angle, args.kind, args.face_rotation) == (1080, 1920, 0.0, "faces", 0.0)
    if side_legs and default_cfg and not args.no_config_legs:
        # (every frame of these legs is checked against the CPU oracle: 64 + 64 1080p frames and 8 4K frames, one host thread each)
        vall = args.verify_frames > 0
        config4_leg = {"upright_faces": config_leg(args, pg, dev, "config-4 leg (upright faces)", 64, 5, 64 if vall else 0, angle=0.8),
                       "rotated_faces": config_leg(args, pg, dev, "config-4 leg (rotated faces)", 64, 5, 64 if vall else 0, angle=0.8, face_rotation=-79.0)}
        config5_leg = config_leg(args, pg, dev, "config-5 leg (4K)", 8, 3, 8 if vall else 0, rows=2160, cols=3840, min_size=20, max_size=2000, shift=0.05,
                                 scale=1.05, det_cap=32768)
        # (8 frames give every XCD ONE frame: the chip is not full.  The same configuration on 64 resident frames -- 531 MB, twice the
        # Infinity Cache -- is the batch the traffic profile profiles/r06_traffic_4k.json was taken on; every frame verified as well)
        config5_leg["batch_of_64"] = config_leg(args, pg, dev, "config-5 leg (4K, 64 frames)", 64, 4, 64 if vall else 0, rows=2160, cols=3840, min_size=20,
                                                max_size=2000, shift=0.05, scale=1.05, det_cap=32768)
        t4 = os.path.join(ROOT, "profiles", "r06_traffic_4k.json")
        if os.path.exists(t4):
            with open(t4) as fh:
                tr4 = json.load(fh)
            config5_leg["batch_of_64"]["roofline"]["traffic"] = int(tr4["fabric_bytes_per_frame"]) * 64
            config5_leg["batch_of_64"]["roofline"]["traffic_source"] = {
                "file": "profiles/r06_traffic_4k.json", "commit": tr4.get("commit"), "frames_per_step_profiled": tr4.get("frames_per_step"),
                "fabric_bytes_per_frame": int(tr4["fabric_bytes_per_frame"]), "ea_read_bytes_per_frame": tr4.get("ea_read_bytes_per_frame_64B_requests"),
                "note": "fabric-side (L2-miss) bytes: 2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes (scripts/gpu_r6_traffic_4k.sh); an upper bound of the HBM bytes"}
        ref_leg = reference_benchmark_leg(args, pg)

    if rank == 0:
        wpf = int(info.windows_per_frame)
        total_frames = n_gpus * B * args.steps
        fps = total_frames / elapsed
        # dominant kernel: the scan.  Variant 2 launches k_scan_tile once per tile class (all classes together
        # read every frame once), variant 1 k_scan_head (+ tail), variant 0 k_scan_mono.
        scan_names = [k for k in ktimes if k.startswith(("scan_", "tail_", "big_"))]
        scan_ms = sum(ktimes[k] for k in scan_names)
        dom = {3: "scan_region+scan_big+big_pool+tail_deep", 2: "scan_tile+tail_deep", 1: "scan_head+scan_tail", 0: "scan_mono"}.get(int(info.variant), "scan")
        alg_bytes = B * args.rows * args.cols + 16 * ndet  # every frame read once + 16 B per emitted detection
        achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms else None
        # Memory traffic of the scan kernels: PMC counters cannot be read live; the committed profile of the same workload
        # (separate rocprofv3 --pmc passes, profiles/rNN_traffic.json) gives FABRIC-side bytes per frame -- what the L2s missed,
        # Infinity-Cache hits included -- scaled here to this batch.  It is an upper bound of the HBM bytes.
        traffic, tnote, tsrc = None, None, None
        tname = {3: "r06_traffic.json", 2: "r01_traffic.json"}.get(int(info.variant))
        tpath = os.path.join(ROOT, "profiles", tname) if tname else None
        for older in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
            if tpath and not os.path.exists(tpath) and int(info.variant) == 3:
                tname = older
                tpath = os.path.join(ROOT, "profiles", tname)
        if tpath and os.path.exists(tpath) and (args.rows, args.cols, args.kind, args.angle) == (1080, 1920, "faces", 0.0):
            with open(tpath) as fh:
                trec = json.load(fh)
            traffic = int(trec.get("fabric_bytes_per_frame", trec.get("hbm_bytes_per_frame"))) * B
            tsrc = {"file": f"profiles/{tname}", "commit": trec.get("commit"), "frames_per_step_profiled": trec.get("frames_per_step"),
                    "fabric_bytes_per_frame": int(trec.get("fabric_bytes_per_frame", 0)),
                    "ea_read_bytes_per_frame": trec.get("ea_read_bytes_per_frame_64B_requests"),
                    "dram_destined_share_of_read_requests": trec.get("dram_destined_share_of_read_requests"),
                    # (per kernel NAME the instantiation with the most requests: k_tail_deep also has a tiny second launch)
                    "l2_hit_rate": {k.split("<")[0]: v.get("hit_rate") for k, v in
                                    sorted((trec.get("l2_per_step") or {}).items(), key=lambda kv: kv[1].get("hits", 0) + kv[1].get("misses", 0))}}
            tnote = (f"profiled offline (profiles/{tname}, commit {trec.get('commit')}): L2-miss (fabric-side) bytes of the scan kernels = 2 x FETCH_SIZE + WRITE_SIZE per frame, "
                     "separate rocprofv3 --pmc passes, x frames; FETCH_SIZE x 2 agrees with TCC_MISS_sum x 128 B on these byte gathers; "
                     f"profiled on {trec.get('frames_per_step')} frames per step; the Infinity Cache sits behind the counted interface (no counter of this stack "
                     "separates its hits), so an upper bound of the HBM bytes")
        out = {
            "metric": "Mwindows/s (1080p facefinder scan, shift 0.1 / scale 1.1)" if (args.rows, args.cols) == (1080, 1920) else "Mwindows/s",
            "value": round(fps * wpf / 1e6, 3),
            "unit": "Mwindows/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8",  # byte compares + float32 leaf sums; not a precision claim
            "data": "synthetic",
            "frames_per_s": round(fps, 2),
            "config": {
                "workload": f"{args.cols}x{args.rows} synthetic gray frames (SYN-{args.kind.upper()}, seed {args.seed}" + (f", faces rotated by {args.face_rotation} deg" if args.face_rotation else "") + "), facefinder cascade, "
                            f"MinSize={args.min_size} MaxSize={args.max_size} Shift={args.shift} Scale={args.scale} angle={args.angle}; "
                            f"{B} HBM-resident frames per GPU per step; RunCascade + per-frame ClusterDetections(iou={args.iou})"
                            + (f" + RCCL all-gather of {gcap}-record cluster lists" if use_dist else ""),
                "frames_per_gpu": B, "windows_per_frame": wpf, "scales": int(info.n_scales), "variant": int(info.variant),
                "head_trees": int(info.n_head_trees), "detections_per_batch": ndet,
                "head_survivor_fraction": round(survivors / (B * wpf), 5) if wpf else None,
                "parallelism": f"frames sharded over {n_gpus} GPU(s), no data-path collective during the scan"
                               + (f"; one all-gather per step via {gather_mode}" if use_dist else ""),
            },
            "verified_frames": verified,
            "verified_frame_indices": verified_idx if verified < B else f"all {B} frames of the timed batch",
            "verification": ("EVERY frame" if verified >= B else "the first and the last frames") +
                            " of the timed batch vs the CPU oracle, raw lists and clusters bit-exact (q 0 ulp); counts.max() <= det_cap"
                            + ("; rank 0's gathered rows vs its lists; one frame of the last rank through the all-gather" if use_dist else ""),
            "gather": gather_mode,
            "kernel_ms_schedule": "per-kernel HIP-event times are taken with the chunked pipeline and the side stream OFF (each launch alone on "
                                  "the stream, as rocprofv3 sees them); the timed step overlaps them, so ms_per_step can be below their sum",
            "kernel_ms": {k: round(v, 4) for k, v in ktimes.items() if k != "end"},
            "cluster_ms": round(cluster_ms, 4) if cluster_ms is not None else None,
            # what running the big scales' launches NEXT to the region launches hides: (sum of the per-kernel times, each launch alone
            # on the stream) + the cluster step - the timed step
            "overlap_ms": round(sum(v for k, v in ktimes.items() if k != "end") + (cluster_ms or 0.0) - elapsed / args.steps * 1e3, 4),
            "roofline": {
                "bound": "hbm", "kernel": "k_" + dom, "kernel_ms_per_batch": round(scan_ms, 4),
                # achieved / frac: algorithmic bytes of a step / the TIMED step (the launches of a step overlap -- the big scales' side
                # chain runs next to the region launches --, so the serial sum of the per-kernel times is not a duration of anything;
                # it stays below as achieved_over_serial_kernel_sum)
                "achieved": round(alg_bytes / (elapsed / args.steps) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                "achieved_over_serial_kernel_sum": round(achieved, 2) if achieved else None,
                # the dominant kernel by itself: k_scan_region's two launches (HIP events, each launch alone on the stream) against the same bytes
                "dominant_kernel": ({"name": "k_scan_region (small + mid group launches)",
                                     "ms_per_step": round(ktimes.get("scan_region_small", 0.0) + ktimes.get("scan_region_mid", 0.0), 4),
                                     "achieved": round(alg_bytes / ((ktimes.get("scan_region_small", 0.0) + ktimes.get("scan_region_mid", 0.0)) * 1e-3) / 1e9, 2),
                                     "frac": round(alg_bytes / ((ktimes.get("scan_region_small", 0.0) + ktimes.get("scan_region_mid", 0.0)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                                    if ktimes.get("scan_region_small") else None),
                "traffic": traffic,
                "traffic_note": tnote,
                "traffic_source": tsrc,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "compulsory bytes only (each frame read once + 16 B per detection); the scan is not bound by HBM: the region kernel keeps its "
                        "LDS pipe ~68 % busy (48 % of those cycles are bank conflicts of divergent byte gathers, as the bank model predicts) and its VALU "
                        "~75-80 % (a wave64 VALU instruction of the kinds the scan is made of costs a SIMD 4 cycles, not 2: profiles/r06_valu_rate.txt, "
                        "r06_experiments.md section 1), the stages behind the first tree are chains of dependent LDS round trips; the 1 % largest windows -- "
                        "gathered from global memory by the side chain that runs NEXT to the region workgroups -- are bound by the L1 fill path (a 128-byte "
                        "line per gathered byte) -- DESIGN.md section 4",
            },
        }
        out["config3_shard"] = shard_leg
        out["config4_rotated"] = config4_leg
        out["config5_4k"] = config5_leg
        out["reference_benchmark"] = ref_leg
        out["single_frame"] = single_leg
        out["gray"] = gray_leg
        out["puploc"] = pup_leg
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, frames, wpf)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
