import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from pigo_amd import core, synth
case = sys.argv[1] if len(sys.argv) > 1 else "sample"
pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())
orc = oracle.OraclePigo.unpack(synth.facefinder_bytes())
if case == "sample":
    img = synth.sample_gray(); rows, cols = 400, 320; args = (int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), 1.1)
elif case == "noise":
    rows, cols = 270, 480; img = synth.syn_noise(rows, cols, seed=99); args = (20, 1000, 0.1, 1.1)
else:
    rows, cols = 270, 480; img = synth.syn_faces(rows, cols, seed=99); args = (20, 1000, 0.1, 1.1)
cp = core.CascadeParams(MinSize=args[0], MaxSize=args[1], ShiftFactor=args[2], ScaleFactor=args[3], ImageParams=core.ImageParams(Pixels=img, Rows=rows, Cols=cols, Dim=cols))
d = pg.RunCascade(cp, 0.0)
w = orc.run_cascade(img, rows, cols, cols, *args, 0.0)
print(case, "ok" if len(d) == len(w) and all((a["row"], a["col"], a["scale"], a["q"]) == (b["row"], b["col"], b["scale"], b["q"]) for a, b in zip(d, w)) else "MISMATCH", len(d), len(w))
