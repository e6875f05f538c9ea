#!/usr/bin/env python3
"""Timeline of ONE k_scan_one launch (plans of a few frames) from the debug library's trace words:

    PIGO_HIP_LIB=pigo_amd/csrc/libpigo_hip_debug.so PIGO_DEBUG_STATS=1 python scripts/one_trace.py [--kind faces] [--rows R --cols C] [VAR=V ...]

Per item class (big bundle / mid region / small region, by the workgroup's first item): when the workgroups finished their first
item, all their items, and when they left (microseconds after the first workgroup started), plus the entries the queues carried."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="faces")
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--shift", type=float, default=0.1)
    ap.add_argument("--scale", type=float, default=1.1)
    ap.add_argument("--angle", type=float, default=0.0)
    ap.add_argument("env", nargs="*")
    a = ap.parse_args()
    os.environ["PIGO_TUNING"] = "1"
    os.environ["PIGO_DEBUG_STATS"] = "1"
    for e in a.env:
        k, v = e.split("=", 1)
        os.environ[k] = v
    import torch
    from pigo_amd import batch, core, synth
    pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    f = synth.make_frames(a.kind, 1, a.rows, a.cols, seed=1234)
    plan = batch.ScanPlan(pg, a.rows, a.cols, ShiftFactor=a.shift, ScaleFactor=a.scale, angle=a.angle, max_frames=1, det_cap=4096)
    dev = torch.from_numpy(f).cuda()
    dets, counts = plan.alloc_outputs(1)
    for _ in range(5):
        plan.run(dev, dets, counts)
    torch.cuda.synchronize()
    plan.status()
    st = plan.debug_stats()
    nreg = max(st[4], 1)
    print("  region phase timers (cycles per region, %d regions x runs): copy %.0f scan(wave 0) %.0f wait %.0f deep(per wave) %.0f total %.0f | deep windows/region %.1f" %
          (st[4], st[0] / nreg, st[1] / nreg, st[3] / nreg, st[2] / nreg / 16, st[5] / nreg, st[7] / nreg))
    print("  queue entries of the last run: %d" % plan.last_queue_count())
    plan.run(dev, dets, counts)  # one more run on zeroed phase timers: the per-workgroup words then belong to ONE item each
    torch.cuda.synchronize()
    tr = plan.debug_trace().astype(np.int64)
    # (the trace words accumulate nothing: every run overwrites them, [6] excepted)
    tr = tr[tr[:, 0] > 0]
    t0 = tr[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    print(f"{a.kind} {a.cols}x{a.rows} angle {a.angle}: {len(tr)} workgroups, {int(counts[0])} detections, entries taken (5 runs) {int(tr[:, 6].sum())}")
    first = tr[:, 5]
    items = tr[:, 4]
    print("  start spread %.1f us; workgroups without an item %d; items per workgroup max %d" % (us(tr[:, 0]).max(), int((items == 0).sum()), int(items.max())))
    for name, col in (("first item done", 1), ("all items done", 2), ("left", 3)):
        v = us(tr[:, col][tr[:, col] > 0])
        if len(v):
            print("  %-16s min %6.1f  median %6.1f  p90 %6.1f  max %6.1f us" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
    order = np.argsort(first)
    srt = tr[order]
    has = srt[:, 4] > 0
    srt = srt[has]
    # item classes by index: thirds of the sorted first-item list are not classes; print the slowest ten items instead
    d = us(srt[:, 1]) - us(srt[:, 0])
    worst = np.argsort(-d)[:10]
    print("  slowest first items: " + ", ".join("item %d %.1f us" % (int(srt[i, 5]), d[i]) for i in worst))
    # per-workgroup region phase timers (cycles, sums over the runs since the plan's debug_stats call: none, the words are copied at
    # the end of each run): copy, scan (wave 0), deep (all waves / 16), wait, per region
    def phases(i):
        w = srt[i, 8:16].astype(np.float64)
        n = max(w[4], 1.0)
        return "copy %5.0f scan %6.0f wait %6.0f deep/wave %6.0f total %6.0f cyc, deep windows %4.0f passes %5.0f" % (w[0] / n, w[1] / n, w[3] / n, w[2] / n / 16, w[5] / n, w[7] / n, w[6] / n)
    small = np.where(srt[:, 5] >= 96)[0]
    if len(small):
        so = small[np.argsort(-d[small])]
        for i in list(so[:6]) + list(so[-4:]):
            print("    item %3d %5.1f us: %s" % (int(srt[i, 5]), d[i], phases(i)))
    for lo, hi in ((0, 32), (32, 96), (96, 1 << 30)):
        m = (srt[:, 5] >= lo) & (srt[:, 5] < hi)
        if m.any():
            print("  first items [%d, %d): duration median %.1f max %.1f us (n=%d)" % (lo, min(hi, int(srt[:, 5].max()) + 1), np.median(d[m]), d[m].max(), int(m.sum())))


if __name__ == "__main__":
    main()
