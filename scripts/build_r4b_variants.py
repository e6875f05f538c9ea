"""Builds the experiment libraries of round 4's second half next to libpigo_hip.so (they travel to the GPU box with the snapshot;
*.so is git-ignored): one per setting of the region kernel's compile-time switches PIGO_OPT_* (pigo_kernels.hip.inc).
    python scripts/build_r4b_variants.py            # all of them, in parallel
scripts/gpu_r4b_ab.sh runs bench.py with PIGO_HIP_LIB pointing at each."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pigo_amd import build  # noqa: E402

SWITCHES = ["DECODE", "LEAFMASK", "L1REG", "COPYX"]
VARIANTS = {
    "base": {},                                   # every switch off: round 4's first-half kernel (the library's default is "all" now)
    "all": {s: 1 for s in SWITCHES},
    "dec": {"DECODE": 1},
    "lm": {"DECODE": 1, "LEAFMASK": 1},
    "l1": {"DECODE": 1, "L1REG": 1},
    "cx": {"COPYX": 1},
    "nol1": {"DECODE": 1, "LEAFMASK": 1, "COPYX": 1},
}


def one(name):
    if name == "debug":  # the debug library (phase timers) of the "all" setting
        return build.build(force=True, debug=True, defines=["PIGO_OPT_%s=1" % s for s in SWITCHES])
    on = VARIANTS[name]
    defines = ["PIGO_OPT_%s=%d" % (s, on.get(s, 0)) for s in SWITCHES]
    out = os.path.join(build.CSRC, "libpigo_hip_x_%s.so" % name)
    return build.build(force=True, defines=defines, out=out)


if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS) + ["debug"]
    with ThreadPoolExecutor(max_workers=8) as ex:
        for path in ex.map(one, names):
            print(path)
