#!/usr/bin/env python3
"""Single-frame numbers for DESIGN.md: (a) pigo_run_cascade on a HOST buffer (PCIe-inclusive: H2D of the frame,
scan, D2H of the detections, two synchronisations) and (b) one HBM-resident frame through the plan API."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pigo_amd import batch, core, synth

pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())
f = synth.make_frames("faces", 1, 1080, 1920, seed=1234)
cp = core.CascadeParams(MinSize=20, MaxSize=1000, ShiftFactor=0.1, ScaleFactor=1.1,
                        ImageParams=core.ImageParams(Pixels=f[0], Rows=1080, Cols=1920, Dim=1920))
for _ in range(3):
    d = pg.RunCascade(cp, 0.0)
t = time.perf_counter()
n = 20
for _ in range(n):
    d = pg.RunCascade(cp, 0.0)
host_ms = (time.perf_counter() - t) / n * 1e3
plan = batch.ScanPlan(pg, 1080, 1920, max_frames=1, det_cap=1024)
dev = torch.from_numpy(f).to("cuda:0")
dets, counts = plan.alloc_outputs(1)
for _ in range(5):
    plan.run(dev, dets, counts)
torch.cuda.synchronize()
t = time.perf_counter()
n = 100
for _ in range(n):
    plan.run(dev, dets, counts)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t) / n * 1e3
w = plan.info().windows_per_frame
print("single 1080p frame: RunCascade(host buffer, PCIe-inclusive) %.3f ms (%.1f Mwindows/s, %d detections); HBM-resident frame, back-to-back %.3f ms (%.1f Mwindows/s)" %
      (host_ms, w / host_ms / 1e3, len(d), dev_ms, w / dev_ms / 1e3))
