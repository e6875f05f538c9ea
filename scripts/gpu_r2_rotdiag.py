"""Which frames overflow a queue in the rotated 1080p batch?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pigo_amd import batch, core, synth
pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())
frames = np.concatenate([synth.make_frames("faces", 34, 1080, 1920, seed=1234), synth.make_frames("noise", 6, 1080, 1920, seed=99)])
d = torch.from_numpy(frames).cuda()
for n0, n1 in ((0, 1), (34, 35), (35, 36), (0, 34), (34, 40), (0, 40)):
    plan = batch.ScanPlan(pg, 1080, 1920, angle=0.8, max_frames=40, det_cap=1024)
    dets, counts = plan.alloc_outputs(n1 - n0)
    plan.run(d[n0:n1], dets, counts)
    torch.cuda.synchronize()
    try:
        plan.status(); print(n0, n1, "ok", int(counts.sum()), plan.last_queue_count())
    except Exception as e:
        print(n0, n1, "ERR", e, plan.last_queue_count())
