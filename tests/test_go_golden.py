"""Reference-pinned parity, for whoever has a Go toolchain: tests/golden/go_golden.json is minted by the REAL package
github.com/esimov/pigo/core (tests/golden/gen_golden.go reading tests/golden/make_go_inputs.py's inputs).  While that file is
absent -- this image has no Go -- the pinned tests skip; the loader itself is exercised against a stand-in of the same schema
minted by the oracle (so the day the real file appears, a failure means a parity gap, not a loader bug).

  CPU:  the C oracle must reproduce every vector of go_golden.json        -> "oracle pinned"
  GPU:  the HIP path must reproduce it through the C ABI                   -> "product pinned"
"""
import json
import os

import numpy as np
import pytest

import oracle
from pigo_amd import core, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "go_golden.json")
INPUTS = os.path.join(ROOT, "tests", "golden", "go_inputs")


def f32(h):
    return np.frombuffer(bytes.fromhex(h), dtype="<f4")[0]


def rows_of(d):
    return [[int(a["row"]), int(a["col"]), int(a["scale"]), np.float32(a["q"]).tobytes().hex()] for a in d]


def load_inputs():
    if not os.path.exists(os.path.join(INPUTS, "manifest.json")):
        import subprocess
        import sys
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_go_inputs.py")], stdout=subprocess.DEVNULL)
    with open(os.path.join(INPUTS, "manifest.json")) as fh:
        return json.load(fh)


def image_of(case):
    return np.fromfile(os.path.join(INPUTS, case["file"]), dtype=np.uint8).reshape(case["rows"], case["dim"])


class OracleEngine:
    """The CPU oracle behind the loader's five calls."""
    make_dets = staticmethod(oracle.make_dets)

    def __init__(self):
        self.pg = oracle.OraclePigo.unpack(synth.facefinder_bytes())

    def run_cascade(self, c, img):
        return self.pg.run_cascade(img, c["rows"], c["cols"], c["dim"], c["min_size"], c["max_size"], c["shift"], c["scale"], c["angle"])

    def cluster(self, dets, iou):
        return self.pg.cluster_detections(dets, iou)  # sorts dets in place like the reference

    def run_detector(self, c, img, rnd):
        o = oracle.OraclePuploc.unpack(synth.cascade_bytes(c["cascade"]))
        return o.run_detector(c["row"], c["col"], float(f32(c["scale"])), c["perturbs"], img, c["rows"], c["cols"], c["dim"], c["angle"], c["flip_v"],
                              rnd, None)

    def gray(self, rgba, kind):
        return oracle.rgb_to_grayscale(rgba, oracle.PIX_RGBA if kind == "RGBA" else oracle.PIX_NRGBA)


class HipEngine:
    """The product through the C ABI."""
    make_dets = staticmethod(core.make_dets)

    def __init__(self):
        self.pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())

    def run_cascade(self, c, img):
        cp = core.CascadeParams(MinSize=c["min_size"], MaxSize=c["max_size"], ShiftFactor=c["shift"], ScaleFactor=c["scale"],
                                ImageParams=core.ImageParams(Pixels=img, Rows=c["rows"], Cols=c["cols"], Dim=c["dim"]))
        return self.pg.RunCascade(cp, c["angle"])

    def cluster(self, dets, iou):
        return self.pg.ClusterDetections(dets, iou)

    def run_detector(self, c, img, rnd):
        plc = core.NewPuplocCascade(0).UnpackCascade(synth.cascade_bytes(c["cascade"]))
        r = plc.RunDetector(core.Puploc(c["row"], c["col"], float(f32(c["scale"])), c["perturbs"]),
                            core.ImageParams(Pixels=img, Rows=c["rows"], Cols=c["cols"], Dim=c["dim"]), c["angle"], c["flip_v"], rnd=rnd)
        return (r.Row, r.Col, np.float32(r.Scale))

    def gray(self, rgba, kind):
        return core.RgbToGrayscale(rgba, core.PIX_RGBA if kind == "RGBA" else core.PIX_NRGBA)


def mint(engine, manifest):
    """What gen_golden.go writes, computed by `engine` (used for the stand-in and, field by field, for the comparison)."""
    out = {"cases": [], "lists": [], "puploc": [], "gray": []}
    for c in manifest["scan"]:
        d = engine.run_cascade(c, image_of(c))
        raw = rows_of(d)
        cl = engine.cluster(d, c["iou"])
        out["cases"].append({"name": c["name"], "detections": raw, "sorted": rows_of(d), "clusters": rows_of(cl)})
    for l in manifest["lists"]:
        d = engine.make_dets([(r, c, s, f32(q)) for r, c, s, q in l["dets"]])
        cl = engine.cluster(d, l["iou"])
        out["lists"].append({"name": l["name"], "sorted": rows_of(d), "clusters": rows_of(cl)})
    return out


def check_against(engine, golden, manifest, q_exact=True):
    """Every vector of `golden` (go_golden.json schema) reproduced by `engine`: integers bit-exact, q bit-exact (the contract
    allows 1e-5; both the oracle and the HIP path target 0 ulp)."""
    def same(got, want, what):
        assert len(got) == len(want), f"{what}: {len(got)} records, Go has {len(want)}"
        for i, (a, b) in enumerate(zip(got, want)):
            assert a[:3] == [int(v) for v in b[:3]], f"{what} record {i}: {a} vs Go {b}"
            if q_exact:
                assert a[3] == b[3], f"{what} record {i}: q {f32(a[3])!r} vs Go {f32(b[3])!r}"
            assert abs(float(f32(a[3])) - float(f32(b[3]))) <= 1e-5, f"{what} record {i}: q outside the 1e-5 contract"

    mine = mint(engine, manifest)
    by_name = {c["name"]: c for c in golden["cases"]}
    for c in mine["cases"]:
        g = by_name[c["name"]]
        same(c["detections"], g["detections"], f"{c['name']} RunCascade")
        same(c["sorted"], g["sorted"], f"{c['name']} sort.Slice order")
        same(c["clusters"], g["clusters"], f"{c['name']} ClusterDetections")
    by_name = {c["name"]: c for c in golden["lists"]}
    for l in mine["lists"]:
        same(l["sorted"], by_name[l["name"]]["sorted"], f"{l['name']} sort.Slice tie order")
        same(l["clusters"], by_name[l["name"]]["clusters"], f"{l['name']} clusters")
    pup = {c["name"]: c for c in manifest["puploc"]}
    for g in golden.get("puploc", []):
        c = pup[g["name"]]
        rnd = np.array([f32(h) for h in g["rnd"]], dtype=np.float32)
        got = engine.run_detector(c, np.fromfile(os.path.join(INPUTS, c["file"]), dtype=np.uint8).reshape(c["rows"], c["dim"]), rnd)
        assert [int(got[0]), int(got[1])] == g["want"][:2] and np.float32(got[2]) == f32(g["want"][2]), (g["name"], got, g["want"])
    gr = {c["name"]: c for c in manifest["gray"]}
    for g in golden.get("gray", []):
        c = gr[g["name"]]
        rgba = np.fromfile(os.path.join(INPUTS, c["file"]), dtype=np.uint8).reshape(c["height"], c["width"], 4)
        assert bytes(engine.gray(rgba, c["kind"])).hex() == g["gray_hex"], g["name"]
    return len(mine["cases"]), len(mine["lists"])


def small(manifest):
    """The stand-in run keeps to the small cases (the 1080p ones cost the CPU suite a second each)."""
    m = dict(manifest)
    m["scan"] = [c for c in manifest["scan"] if c["rows"] * c["cols"] <= 480 * 640][:6]
    m["lists"] = manifest["lists"][:5]
    return m


def test_loader_and_schema_against_an_oracle_minted_stand_in():
    """Not a parity claim: proves that the loader, the schema and the comparison code work, so that go_golden.json can be
    dropped in without touching the tests.  A deliberately corrupted copy must be caught."""
    manifest = small(load_inputs())
    eng = OracleEngine()
    stand_in = mint(eng, manifest)
    stand_in["source"] = "oracle stand-in (NOT the Go package)"
    assert check_against(eng, stand_in, manifest) == (len(manifest["scan"]), len(manifest["lists"]))
    bad = json.loads(json.dumps(stand_in))
    victim = next(c for c in bad["cases"] if c["detections"])
    victim["detections"][0][1] += 1
    with pytest.raises(AssertionError, match="RunCascade"):
        check_against(eng, bad, manifest)
    bad = json.loads(json.dumps(stand_in))
    tied = next(l for l in bad["lists"] if len(l["sorted"]) > 12)
    tied["sorted"][0], tied["sorted"][1] = tied["sorted"][1], tied["sorted"][0]
    with pytest.raises(AssertionError, match="tie order"):
        check_against(eng, bad, manifest)


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="tests/golden/go_golden.json not minted (no Go toolchain here): parity stays unpinned")
def test_oracle_reproduces_the_go_reference():
    with open(GOLDEN) as fh:
        golden = json.load(fh)
    assert "esimov/pigo" in golden.get("source", "")
    check_against(OracleEngine(), golden, load_inputs())


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="tests/golden/go_golden.json not minted (no Go toolchain here): parity stays unpinned")
def test_hip_path_reproduces_the_go_reference():
    with open(GOLDEN) as fh:
        golden = json.load(fh)
    check_against(HipEngine(), golden, load_inputs())


@pytest.mark.gpu
def test_hip_path_against_the_stand_in_inputs():
    """The same loader driven by the product on the go_inputs (incl. the 1080p cases and every tie list), against the
    oracle: what test_hip_path_reproduces_the_go_reference will do once go_golden.json exists."""
    manifest = load_inputs()
    stand_in = mint(OracleEngine(), manifest)
    check_against(HipEngine(), stand_in, manifest)
