#!/usr/bin/env python3
"""bench.py -- the 1080p facefinder scan benchmark (BASELINE.json metric: Mwindows/s and frames/s; % HBM roofline).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One STEP = one pass of the hot path over one batch of synthetic 1080p frames that are already resident in
HBM: RunCascade (scan + order restore) for every frame, per-frame ClusterDetections on the GPU, and -- for
N > 1 -- one RCCL all-gather of the fixed-capacity per-frame cluster lists.  Workload = BASELINE.json
configs[1] (1920x1080, facefinder, MinSize 20, MaxSize 1000, ShiftFactor 0.1, ScaleFactor 1.1) applied to a
batch of `--frames` seeded SYN-FACES frames per GPU (weak scaling: per-GPU work is fixed as N grows).

Rank 0 prints ONE JSON line (see the task contract) with two extra objects:
  roofline      HBM roofline of the dominant kernel (k_scan_head): algorithmic bytes per launch (every frame
                read once + 16 B per detection) / that kernel's mean duration measured with HIP events on the
                launch stream; peak 8.0 TB/s.
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
    ap.add_argument("--variant", type=int, default=None, help="0 = monolithic, 1 = head+queue+tail, 2 = LDS tile (default)")
    ap.add_argument("--no-cluster", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even for one rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gray", action="store_true", help="skip the side measurements (RgbToGrayscale, RunDetector, single frame)")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the one-frame-per-call leg (keeps kernel profiles per-batch)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU sample (0 = auto, ~15 s)")
    return ap.parse_args()


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


def main():
    args = parse_args()
    # stdout carries exactly ONE line, the JSON record: native libraries (RCCL prints its version banner to stdout) and
    # anything else that writes to fd 1 go to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    n_gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B = args.frames
    frames = synth.make_frames(args.kind, B, args.rows, args.cols, seed=args.seed, first_index=rank * B)
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

    def step():
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
    plan.status()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- per-kernel timing (HIP events on the launch stream), outside the timed region ----
    plan.set_profiling(True)
    ktimes = {}
    reps = 5
    for _ in range(reps):
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

    # ---- BASELINE configs[1] as stated: ONE 1080p frame.  (a) resident in HBM, back-to-back launches of the plan;
    # (b) RunCascade on a host buffer: H2D of the frame, scan, D2H of the detections (PCIe-inclusive; never `value`)
    single_leg = None
    if rank == 0 and not args.no_gray and not args.no_single_frame:
        plan1 = batch.ScanPlan(pg, args.rows, args.cols, MinSize=args.min_size, MaxSize=args.max_size, ShiftFactor=args.shift,
                               ScaleFactor=args.scale, angle=args.angle, max_frames=1, det_cap=args.det_cap)
        d1, c1 = plan1.alloc_outputs(1)
        for _ in range(5):
            plan1.run(d_frames[:1], d1, c1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(100):
            plan1.run(d_frames[:1], d1, c1)
        torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t1) / 100 * 1e3
        plan1.status()
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
                      "host_buffer_ms": round(host_ms, 4), "host_buffer_mwindows_per_s": round(w1 / host_ms / 1e3, 1),
                      "note": "one frame per call; host_buffer includes PCIe H2D/D2H and two synchronisations"}
        del plan1

    # ---- side measurement, outside the timed region: RgbToGrayscale (core/grayscale.go:8-23), the streaming step in
    # front of the scan.  RGBA frames {g,g,g,255} built on the GPU from the gray batch; the kernel must give them back.
    gray_leg = None
    if rank == 0 and not args.no_gray:
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
    if rank == 0 and not args.no_gray:
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
        pup_leg = {"kernel": "k_puploc", "requests": nreq, "perturbs": 63, "ms_per_launch": round(pms, 4),
                   "requests_per_s": round(nreq / (pms * 1e-3), 1), "tree_walks_per_s": round(nreq * 63 * 100 / (pms * 1e-3), 1)}
        if not args.no_cpu_baseline:
            import oracle
            oplc = oracle.OraclePuploc.unpack(synth.cascade_bytes("puploc"))
            t = time.perf_counter()
            for i in range(40):
                r = reqs[i]
                oplc.run_detector(int(r["row"]), int(r["col"]), float(r["scale"]), 63, frames[r["frame"]], args.rows, args.cols, args.cols, 0.0,
                                  False, rnd[i], None)
            pup_leg["cpu_requests_per_s_one_thread"] = round(40 / (time.perf_counter() - t), 1)

    if rank == 0:
        wpf = int(info.windows_per_frame)
        total_frames = n_gpus * B * args.steps
        fps = total_frames / elapsed
        # dominant kernel: the scan.  Variant 2 launches k_scan_tile once per tile class (all classes together
        # read every frame once), variant 1 k_scan_head (+ tail), variant 0 k_scan_mono.
        scan_names = [k for k in ktimes if k.startswith("scan_") or k.startswith("tail_")]
        scan_ms = sum(ktimes[k] for k in scan_names)
        dom = {2: "scan_tile+tail_deep", 1: "scan_head+scan_tail", 0: "scan_mono"}.get(int(info.variant), "scan")
        alg_bytes = B * args.rows * args.cols + 16 * ndet  # every frame read once + 16 B per emitted detection
        achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms else None
        # HBM traffic: PMC counters cannot be read live; the committed profile of the same workload (separate rocprofv3
        # --pmc passes, profiles/r01_traffic.json) gives bytes per frame, scaled here to this batch
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and (args.rows, args.cols, args.kind, args.angle) == (1080, 1920, "faces", 0.0) and int(info.variant) == 2:
            with open(tpath) as fh:
                traffic = int(json.load(fh)["hbm_bytes_per_frame"]) * B
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
                "workload": f"{args.cols}x{args.rows} synthetic gray frames (SYN-{args.kind.upper()}, seed {args.seed}), facefinder cascade, "
                            f"MinSize={args.min_size} MaxSize={args.max_size} Shift={args.shift} Scale={args.scale} angle={args.angle}; "
                            f"{B} HBM-resident frames per GPU per step; RunCascade + per-frame ClusterDetections(iou={args.iou})"
                            + (f" + RCCL all-gather of {gcap}-record cluster lists" if use_dist else ""),
                "frames_per_gpu": B, "windows_per_frame": wpf, "scales": int(info.n_scales), "variant": int(info.variant),
                "head_trees": int(info.n_head_trees), "detections_per_batch": ndet,
                "head_survivor_fraction": round(survivors / (B * wpf), 5) if wpf else None,
                "parallelism": f"frames sharded over {n_gpus} GPU(s), no data-path collective during the scan",
            },
            "kernel_ms": {k: round(v, 4) for k, v in ktimes.items() if k != "end"},
            "cluster_ms": round(cluster_ms, 4) if cluster_ms is not None else None,
            "roofline": {
                "bound": "hbm", "kernel": "k_" + dom, "kernel_ms_per_batch": round(scan_ms, 4),
                "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
                "traffic": traffic,
                "traffic_note": "profiled offline (profiles/r01_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE per frame x frames); dominated by the deep tail's footprint copies" if traffic else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "compulsory bytes only (each frame read once); the kernel is gather/issue bound, see DESIGN.md",
            },
        }
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
