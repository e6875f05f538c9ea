"""In-process A/B of plan settings on bench.py's default workload (128 resident 1080p SYN-FACES frames, RunCascade +
ClusterDetections per step), every setting checked against the CPU oracle on the first two and the last two frames of the batch
(raw lists and clusters, bit-exact) -- the frames and the oracle's answers are made once per process.

    PIGO_HIP_LIB=pigo_amd/csrc/libpigo_hip_x_all.so python scripts/ab.py "name:VAR=V VAR2=V" "other:VAR=W" ...

A spec's variables are set (under PIGO_TUNING=1) while its plan is built -- the library reads its tuning switches at plan creation --
and removed afterwards.  Optional arguments before the specs: --frames N --steps K --reps R --angle A --kernel-times.
Prints one line per spec: name, ms per step (best of R repetitions of K steps), Gwindows/s, verification result.
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--angle", type=float, default=0.0)
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--min-size", type=int, default=20)
    ap.add_argument("--max-size", type=int, default=1000)
    ap.add_argument("--shift", type=float, default=0.1)
    ap.add_argument("--scale", type=float, default=1.1)
    ap.add_argument("--det-cap", type=int, default=1024)
    ap.add_argument("--face-rotation", type=float, default=0.0)
    ap.add_argument("--kernel-times", action="store_true")
    ap.add_argument("--no-cluster", action="store_true", help="the step is RunCascade alone (bench.py's single_frame leg)")
    ap.add_argument("--kind", default="faces")
    ap.add_argument("specs", nargs="+")
    a = ap.parse_args()
    os.environ["PIGO_TUNING"] = "1"
    import torch
    import oracle
    from pigo_amd import batch, core, synth

    n = a.frames
    frames = synth.make_frames(a.kind, n, a.rows, a.cols, seed=1234, rotate_deg=a.face_rotation)
    idx = sorted(set([0, 1, n - 2, n - 1]) & set(range(n)))
    orc = oracle.OraclePigo.unpack(synth.facefinder_bytes())
    want, wantc = {}, {}

    def work(f):
        want[f] = orc.run_cascade(frames[f], a.rows, a.cols, a.cols, a.min_size, a.max_size, a.shift, a.scale, a.angle)
        wantc[f] = orc.cluster_detections(want[f].copy(), 0.2)

    th = [threading.Thread(target=work, args=(f,)) for f in idx]
    for t in th:
        t.start()
    d_frames = torch.from_numpy(frames).cuda()
    pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    for t in th:
        t.join()
    lib = os.environ.get("PIGO_HIP_LIB", "default")
    print(f"# lib {os.path.basename(lib)}  frames {n}  steps {a.steps} x {a.reps}  angle {a.angle}", flush=True)

    def same(x, y):
        if len(x) != len(y):
            return f"{len(x)} records vs oracle {len(y)}"
        for i in range(len(x)):
            if (int(x[i]["row"]), int(x[i]["col"]), int(x[i]["scale"])) != (int(y[i]["row"]), int(y[i]["col"]), int(y[i]["scale"])) or \
                    np.float32(x[i]["q"]) != np.float32(y[i]["q"]):
                return f"record {i}: {x[i]} vs oracle {y[i]}"
        return None

    for spec in a.specs:
        name, _, envs = spec.partition(":")
        kv = dict(e.split("=", 1) for e in envs.split() if "=" in e)
        for k, v in kv.items():
            os.environ[k] = v
        try:
            pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())  # (a handle per spec: some switches are read when the cascade is unpacked)
            plan = batch.ScanPlan(pg, a.rows, a.cols, MinSize=a.min_size, MaxSize=a.max_size, ShiftFactor=a.shift, ScaleFactor=a.scale,
                                  angle=a.angle, max_frames=n, det_cap=a.det_cap)
        finally:
            for k in kv:
                os.environ.pop(k, None)
        dets, counts = plan.alloc_outputs(n)
        cl = plan.alloc_cluster_outputs(dets, counts)

        def step():
            plan.run(d_frames, dets, counts)
            if not a.no_cluster:
                plan.cluster(dets, counts, 0.2, out=cl)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        plan.status()
        best = 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / a.steps * 1e3)
        plan.status()
        got = batch.dets_to_numpy(dets, counts)
        gcl = batch.dets_to_numpy(cl[1], cl[2])
        bad = None
        for f in idx:
            bad = bad or same(got[f], want[f]) or (None if a.no_cluster else same(gcl[f], wantc[f]))
        kt = ""
        if a.kernel_times:
            plan.set_profiling(True)
            plan.run(d_frames, dets, counts)
            torch.cuda.synchronize()
            kt = " " + " ".join(f"{k}={v:.3f}" for k, v in plan.last_timings())
            plan.set_profiling(False)
        wpf = int(plan.info().windows_per_frame)
        print(f"{name:28s} {best:8.4f} ms  {n * wpf / best / 1e6:7.2f} Gwin/s  variant {int(plan.info().variant)}  "
              f"{'VERIFIED' if bad is None else 'MISMATCH ' + bad}{kt}", flush=True)
        del plan, dets, counts, cl


if __name__ == "__main__":
    main()
