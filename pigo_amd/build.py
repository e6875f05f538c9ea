"""Builds libpigo_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

The built .so stays next to its sources (pigo_amd/csrc/) so that it travels to the GPU box with the
repo snapshot and shows up as in-tree native code in the driver's loaded-library report.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libpigo_hip.so")
#: debug build (-DPIGO_DEBUG_BUILD): phase timers inside the scan kernels (PIGO_DEBUG_STATS=1) and the superseded variant-1
#: kernels; never loaded unless PIGO_HIP_LIB points at it (scripts/ A/B runs)
LIB_DEBUG = os.path.join(CSRC, "libpigo_hip_debug.so")
SOURCES = [os.path.join(CSRC, "pigo_hip.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "pigo_kernels.hip.inc"), os.path.join(HERE, "..", "include", "pigo_hip.h")]

# -ffp-contract=off: the float32 leaf sums and the float64 IoU must be evaluated exactly as written
# (DESIGN.md "Bit-exactness"); gfx950 is the only target -- no other arch, no host fallback.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wextra"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm): libpigo_hip.so cannot be built")


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, debug=False, defines=(), out=None):
    """Compile libpigo_hip.so (or, with debug=True, libpigo_hip_debug.so) if it is missing or older than its sources.
    `defines` / `out` build experimental variants next to it (scripts/ A/B runs).  Returns the path."""
    lib = out or (LIB_DEBUG if debug else LIB)
    if force or needs_build(lib):
        flags = list(HIPCC_FLAGS) + (["-DPIGO_DEBUG_BUILD"] if debug else []) + ["-D" + d for d in defines]
        cmd = [hipcc()] + flags + SOURCES + ["-o", lib + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, debug="--debug" in sys.argv[1:]))
