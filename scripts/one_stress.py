#!/usr/bin/env python3
"""Stress of k_scan_one's in-launch hand-offs (plans of a few frames): thousands of launches, every result compared with the
first (itself checked against the CPU oracle), alone and NEXT TO a live 64-frame plan running on another stream (uneven load:
the case in which a hand-off bug shows; MI355X_MICROARCH.md "Test every hand-off under UNEVEN load").

    python scripts/one_stress.py [--launches N] [--sizes 1080x1920,400x320,...] [--frames F] [VAR=V ...]

Prints one line per (size, mode) with the launches, mismatches, status failures and ms per launch; exit code 1 on any failure."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=4000)
    ap.add_argument("--sizes", default="1080x1920,400x320,720x1280")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--angle", type=float, default=0.0)
    ap.add_argument("--kind", default="faces")
    ap.add_argument("env", nargs="*")
    a = ap.parse_args()
    if a.env:
        os.environ["PIGO_TUNING"] = "1"
    for e in a.env:
        k, v = e.split("=", 1)
        os.environ[k] = v
    import torch
    import oracle
    from pigo_amd import batch, core, synth

    pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    orc = oracle.OraclePigo.unpack(synth.facefinder_bytes())
    bad_total = 0
    for size in a.sizes.split(","):
        rows, cols = (int(v) for v in size.split("x"))
        f = synth.make_frames(a.kind, a.frames, rows, cols, seed=77)
        plan = batch.ScanPlan(pg, rows, cols, angle=a.angle, max_frames=a.frames, det_cap=4096)
        assert plan.info().variant == 3, plan.info().variant
        dev = torch.from_numpy(f).cuda()
        dets, counts = plan.alloc_outputs(a.frames)
        plan.run(dev, dets, counts)
        torch.cuda.synchronize()
        plan.status()
        ref_d, ref_c = dets.clone(), counts.clone()
        got = batch.dets_to_numpy(dets, counts)
        for fi in range(a.frames):
            want = orc.run_cascade(f[fi], rows, cols, cols, 20, 1000, 0.1, 1.1, a.angle)
            assert len(got[fi]) == len(want), (len(got[fi]), len(want))
            for g, w in zip(got[fi], want):
                assert (g["row"], g["col"], g["scale"], g["q"]) == (w["row"], w["col"], w["scale"], w["q"]), (g, w)
        # the neighbour: a 64-frame 1080p plan kept busy on its own stream
        big = batch.ScanPlan(pg, 1080, 1920, max_frames=64, det_cap=1024)
        bf = torch.from_numpy(synth.make_frames("faces", 64, 1080, 1920, seed=5)).cuda()
        bd, bc = big.alloc_outputs(64)
        side = torch.cuda.Stream()
        for mode in ("alone", "next to a 64-frame plan"):
            bad = fails = 0
            t0 = time.perf_counter()
            for i in range(a.launches):
                if mode != "alone" and i % 16 == 0:
                    with torch.cuda.stream(side):
                        big.run(bf, bd, bc, stream=side)
                dets.zero_()
                plan.run(dev, dets, counts)
                if i % 8 == 7 or mode != "alone":
                    torch.cuda.synchronize() if mode == "alone" else torch.cuda.current_stream().synchronize()
                    try:
                        plan.status()
                    except core.PigoError as e:
                        fails += 1
                        if fails < 3:
                            print("   ", e)
                    if not (torch.equal(counts, ref_c) and torch.equal(dets, ref_d)):
                        bad += 1
                        if bad < 3:
                            nd = int((dets != ref_d).any(dim=2).sum())
                            print("    launch %d: counts %s (want %s), %d records differ" % (i, counts.tolist(), ref_c.tolist(), nd))
            torch.cuda.synchronize()
            big.status()
            ms = (time.perf_counter() - t0) / a.launches * 1e3
            print("%-10s x%d %-26s launches %6d  mismatches %d  status failures %d  %.4f ms / launch (checks included)" % (size, a.frames, mode, a.launches, bad, fails, ms))
            bad_total += bad + fails
        del big
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
