import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from pigo_amd import batch, core, synth

orc = oracle.OraclePigo.unpack(synth.facefinder_bytes())
rng = np.random.default_rng(2024 + 2)
cases = []
for k in range(14):
    rows, cols = int(rng.integers(30, 360)), int(rng.integers(30, 480))
    dim = cols + int(rng.integers(0, 17))
    img = np.full((rows, dim), 200, dtype=np.uint8)
    kind = synth.syn_faces if k % 3 else synth.syn_noise
    img[:, :cols] = kind(rows, cols, seed=31, frame_index=k)
    mn, mx = int(rng.integers(0, 60)), int(rng.integers(20, 700))
    shift = float(rng.choice([0.02, 0.05, 0.1, 0.15, 0.2, 0.5]))
    scale = float(rng.choice([1.0, 1.03, 1.05, 1.1, 1.15, 1.3, 2.0]))
    angle = float(rng.choice([0.0, 0.0, 0.0, 0.03, 0.125, 0.5, 0.8, 1.0, 1.7, -0.3]))
    iou = float(rng.choice([0.0, 0.01, 0.1, 0.15, 0.2]))
    cases.append((k, rows, cols, dim, img, mn, mx, shift, scale, angle))
pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())
for (k, rows, cols, dim, img, mn, mx, shift, scale, angle) in cases:
    if k not in (7, 10): continue
    if os.environ.get('FORCE_ANGLE'): angle = float(os.environ['FORCE_ANGLE'])
    want = orc.run_cascade(img, rows, cols, dim, mn, mx, shift, scale, angle)
    d = torch.from_numpy(img[None].copy()).to("cuda:0")
    res = {}
    for variant in (2, 0):
        plan = batch.ScanPlan(pg, rows, cols, dim, MinSize=mn, MaxSize=mx, ShiftFactor=shift, ScaleFactor=scale, angle=angle, max_frames=1, det_cap=4096)
        plan.set_variant(variant)
        dets, counts = plan.alloc_outputs(1)
        plan.run(d, dets, counts)
        torch.cuda.synchronize()
        try:
            plan.status(); st = "ok"
        except Exception as e:
            st = str(e)[:60]
        got = batch.dets_to_numpy(dets, counts, 0)
        if variant == 2 and os.environ.get("PIGO_DEBUG_STATS") and k == 7:
            import ctypes as C
            buf = (C.c_uint64 * (16 * 256))()
            plan.L.pigo_plan_debug_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
            rc = plan.L.pigo_plan_debug_trace(plan._h, buf, 16 * 256)
            for b in range(16):
                rec = buf[b * 256:(b + 1) * 256]
                its = rec[0]
                if its == 0: continue
                print("  tile", b, "iters", its)
                for it in range(min(its, 14)):
                    a0, a1, okm, a3 = rec[1 + it * 4: 5 + it * 4]
                    thr = np.array([a3 & 0xffffffff], dtype=np.uint32).view(np.float32)[0]
                    print("     k0 %d kend %d tab [%d,%d) alive %d n %d thr[k0] %.4f" % (a0 >> 32, a0 & 0xffffffff, a1 >> 32, a1 & 0xffffffff, bin(okm).count("1"), a3 >> 32, thr))
        res[variant] = (len(got), st, plan.last_queue_count())
        if variant == 2 and len(got) != len(want):
            ws = {(int(a["row"]), int(a["col"]), int(a["scale"])) for a in want}
            extra = [(int(a["row"]), int(a["col"]), int(a["scale"]), float(a["q"])) for a in got if (int(a["row"]), int(a["col"]), int(a["scale"])) not in ws]
            print("   extra:", extra[:12])
    print(k, f"{rows}x{cols} dim{dim} [{mn},{mx}] {shift}/{scale} a={angle}", "want", len(want), "v2", res[2], "v0", res[0], "windows", plan.info().windows_per_frame, "scales", plan.info().n_scales)
