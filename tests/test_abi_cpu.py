"""CPU tests of the drop-in boundary: the C-ABI library is built, loads, exports every symbol the header
declares, reports errors instead of aborting -- and its host-side pieces (Go's sort.Slice restated, the
scale ladder) agree with the oracle.  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from pigo_amd import core, distributed, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pigo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # strip comments
    declared = sorted(set(re.findall(r"\b(pigo_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 18
    L = core.load_library()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(core.ABI_SYMBOLS) == declared  # the Python mirror binds exactly the header's surface


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(core, "_lib", None)
    monkeypatch.setattr(core._build, "LIB", "/nonexistent/libpigo_hip.so")
    with pytest.raises(core.PigoError, match="no CPU fallback"):
        core.load_library()


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under pigo_amd/ may import, link or execute it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pigo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "libpigo_oracle" not in text, f


def test_unpack_error_codes_without_gpu(packet):
    """Packet validation happens before any device call, so it is testable on CPU: the reference would
    panic on a short packet (pigo.go:64,81,90) -> PIGO_ERR_PACKET -> PigoPanic."""
    pg = core.NewPigo(0)
    for cut in (0, 7, 15, 16 + 100, len(packet) - 1):
        with pytest.raises(core.PigoPanic):
            pg.Unpack(packet[:cut])
    bad = bytearray(packet)
    bad[8:12] = (40).to_bytes(4, "little")  # absurd tree depth
    with pytest.raises((ValueError, core.PigoPanic)):
        pg.Unpack(bytes(bad))
    L = core.load_library()
    if L.pigo_device_count() == 0:  # no GPU here: a good packet must fail with a HIP error, never fall back
        with pytest.raises(core.PigoError):
            pg.Unpack(packet)


def test_host_go_sort_matches_oracle_restatement_including_ties():
    """pigo_sort_by_q (product, C++) and oracle_sort_by_q (oracle, C) restate the same Go algorithm; they must
    produce the same permutation on tie-heavy inputs where an unstable sort is observable."""
    rng = np.random.default_rng(11)
    for n in (0, 1, 2, 12, 13, 20, 49, 50, 51, 64, 200, 317, 1000, 13120):
        for levels in (3, 17, 10**6):
            q = (rng.integers(1, levels + 1, n) / np.float32(7.0)).astype(np.float32)
            a = core.make_dets([(i, 2 * i, 3 * i, q[i]) for i in range(n)])
            b = oracle.make_dets([(i, 2 * i, 3 * i, q[i]) for i in range(n)])
            core.sort_by_q(a)
            oracle.sort_by_q(b)
            assert (a["row"] == b["row"]).all() and (a["q"] == b["q"]).all(), (n, levels)
            assert (np.diff(a["q"]) >= 0).all()
    for pat in (np.arange(500), np.arange(500)[::-1], np.arange(500) % 2, np.concatenate([np.arange(250), np.arange(250)[::-1]])):
        a = core.make_dets([(i, 0, 0, float(v)) for i, v in enumerate(pat)])
        b = oracle.make_dets([(i, 0, 0, float(v)) for i, v in enumerate(pat)])
        core.sort_by_q(a)
        oracle.sort_by_q(b)
        assert (a["row"] == b["row"]).all()


def test_shard_bounds_cover_the_batch():
    for n in (0, 1, 7, 8, 9, 8192, 1000):
        for world in (1, 2, 3, 4, 8):
            spans = [distributed.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
            idx, per = distributed.gathered_frame_index(n, world)
            assert sorted(i for i in idx.tolist() if i >= 0) == list(range(n)) and per * world == len(idx)


def test_synthetic_frames_are_deterministic():
    a = synth.syn_faces(240, 320, seed=1234, frame_index=5)
    b = synth.syn_faces(240, 320, seed=1234, frame_index=5)
    c = synth.syn_faces(240, 320, seed=1234, frame_index=6)
    assert (a == b).all() and (a != c).any()
    n = synth.syn_noise(64, 64, seed=1, frame_index=0)
    assert n.dtype == np.uint8 and 100 < n.mean() < 155


def test_cpp_mirror_compiles_and_links(tmp_path):
    """include/pigo.hpp (the C++ mirror of the Go API) compiles against the header, links against the library and,
    without a GPU, reports the missing device as an exception (exit 4) instead of aborting or falling back."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    core.load_library()
    exe = str(tmp_path / "cpp_mirror_check")
    csrc = os.path.join(ROOT, "pigo_amd", "csrc")
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp_mirror_check.cpp"), "-o", exe, "-L" + csrc, "-lpigo_hip", "-Wl,-rpath," + csrc])
    r = subprocess.run([exe, os.path.join(ROOT, "pigo_amd", "data", "facefinder"), os.path.join(ROOT, "pigo_amd", "data", "sample_gray_320x400.bin")],
                       capture_output=True, text=True)
    if core.load_library().pigo_device_count() == 0:
        assert r.returncode == 4 and "error:" in r.stdout, (r.returncode, r.stdout, r.stderr)
    else:
        assert r.returncode == 0 and "clusters=1 (206,154,261," in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_pruned_go_sort_equals_plain_go_sort_on_the_host(tmp_path):
    """k_gosort_ties does not sort the parts of a list that hold no equal Q values (gosort::KeyData::prune).  The same
    host/device code, compiled for the host only, must give exactly what the plain restatement of Go's pdqsort gives on 600
    random lists with ties of every density (tests/gosort_prune_check.hip).  core/pigo.go:264-266."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "gosort_prune_check")
    subprocess.check_call([hipcc, "--cuda-host-only", "-O1", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "gosort_prune_check.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok lists=600"), (r.returncode, r.stdout, r.stderr)
    assert int(r.stdout.split("kept_from_stable=")[1]) > 10000  # the pruning did happen


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """libpigo_hip.so and PyTorch must share ONE libamdhip64 / libhsa-runtime64: with two HSA runtimes in a process the
    second one finds no GPUs (seen on the GPU box when pigo_amd was loaded before torch)."""
    import subprocess
    import sys
    prog = ("import sys; sys.path.insert(0, %r)\n"
            "%s\n"
            "libs = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime64' in l}\n"
            "print(len([l for l in libs if 'libamdhip64' in l]), len([l for l in libs if 'libhsa-runtime64' in l]))\n")
    orders = ["from pigo_amd import core; core.load_library(); import torch",
              "import torch; from pigo_amd import core; core.load_library()",
              "from pigo_amd import core; core.load_library()"]
    for o in orders:
        r = subprocess.run([sys.executable, "-c", prog % (ROOT, o)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        assert r.stdout.split() == ["1", "1"], (o, r.stdout)
