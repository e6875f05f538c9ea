"""What a batch plan alive in the process does to a one-frame plan's calls (hardware queues are shared by the process's streams):
    python scripts/single_in_process.py [nbatch_plans]
times 100 back-to-back runs of a one-frame plan alone, then again after creating (and running once) `nbatch_plans` 128-frame plans."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pigo_amd import batch, core, synth

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())
f = synth.make_frames("faces", 8, 1080, 1920, seed=1234)
dev = torch.from_numpy(f).to("cuda:0")


def single(tag):
    plan = batch.ScanPlan(pg, 1080, 1920, max_frames=1, det_cap=1024)
    dets, counts = plan.alloc_outputs(1)
    for _ in range(5):
        plan.run(dev[:1], dets, counts)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(100):
            plan.run(dev[:1], dets, counts)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / 100 * 1e3)
    print(f"{tag}: one-frame plan {best:.4f} ms per call", flush=True)
    return plan


p0 = single("alone")
del p0
keep = []
for i in range(nb):
    pl = batch.ScanPlan(pg, 1080, 1920, max_frames=8, det_cap=1024)
    d, c = pl.alloc_outputs(8)
    pl.run(dev, d, c)
    torch.cuda.synchronize()
    keep.append((pl, d, c))
    single(f"with {i + 1} batch plan(s) alive")
