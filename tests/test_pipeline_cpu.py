"""CPU tests of the result plumbing (scope row f4, pigo_amd/pipeline.py): the CLI's JSON schema with its quirks
(cmd/pigo/main.go:88-100,224,358-578) on a scripted engine, and the whole flow on the oracle-backed engine."""
import itertools
import json

import numpy as np
import pytest

from pigo_amd import core, pipeline, synth

from oracle_engine import OracleEngine


class ScriptedEngine:
    """Returns canned results so that the expected JSON can be written by hand."""

    def __init__(self, faces, pups):
        self.faces, self.pups = faces, iter(pups)
        self.calls = []

    def unpack(self, packet):
        return "classifier"

    def unpack_puploc(self, packet):
        return "plc:%d" % len(packet)

    def rgb_to_grayscale(self, rgba, kind):
        return np.zeros(rgba.shape[0] * rgba.shape[1], np.uint8)

    def run_cascade(self, *a):
        return self.faces

    def cluster_detections(self, classifier, dets, iou):
        return dets

    def run_detector(self, plc, pl, pixels, rows, cols, dim, angle, flip_v, rnd, pool):
        self.calls.append(("run", plc, pl.Row, pl.Col, pl.Scale, pl.Perturbs, flip_v, len(rnd)))
        return next(self.pups)

    def get_landmark_point(self, flpc, left, right, pixels, rows, cols, dim, perturb, flip_v, rnd, pool):
        self.calls.append(("flp", flpc, left.Row, right.Row, perturb, flip_v))
        return next(self.pups)


def test_json_schema_quirks_on_a_scripted_engine():
    faces = core.make_dets([(100, 50, 60, 10.0), (200, 200, 80, 3.0), (30, 40, 60, 6.0), (300, 300, 40, 9.0)])
    P = core.Puploc
    pups = [P(95, 40, 14.9), P(96, 0, 15.0),          # face 0: left eye kept, right eye dropped (Col == 0)
            P(25, 33, 15.5), P(26, 47, 15.25)]        # face 2: both eyes
    eng = ScriptedEngine(faces, pups)
    fd = pipeline.FaceDetector(cascade=b"c", puploc=b"pp", engine=eng, state=pipeline.DetectorState(lambda n: np.zeros(n, np.float32)))
    got = fd.detect_json(np.zeros((400, 400, 4), np.uint8))
    want = ('[{"eyes":[{"x":40,"y":95,"size":14}],"face":{"x":20,"y":70,"size":60}},'
            # face 1 has Q <= 5: skipped.  Face 2: "y" is 30 - 60/2 = 0 -> omitted; its eyes list still holds face 0's eye
            '{"eyes":[{"x":40,"y":95,"size":14},{"x":33,"y":25,"size":15},{"x":47,"y":26,"size":15}],"face":{"x":10,"size":60}},'
            # face 3: Scale <= 50 -> no eye search, but the cumulative list is attached all the same
            '{"eyes":[{"x":40,"y":95,"size":14},{"x":33,"y":25,"size":15},{"x":47,"y":26,"size":15}],"face":{"x":280,"y":280,"size":40}}]\n')
    assert got == want
    json.loads(got)
    # the eye requests of main.go:415-461: 63 perturbations, float32 offsets
    assert eng.calls[0] == ("run", "plc:2", 100 - 4, 50 - 10, 15.0, 63, False, 189)
    assert eng.calls[1][:5] == ("run", "plc:2", 96, 50 + 11, 15.0)
    assert pipeline.encode_json([]) == "[]\n" and fd.draw_faces(core.make_dets([(1, 1, 99, 5.0)])) == []  # Q > 5.0 is strict


def test_landmark_call_order_and_requirements():
    faces = core.make_dets([(100, 100, 120, 9.0)])
    P = core.Puploc
    pups = [P(90, 80, 30.0), P(90, 120, 30.0)] + [P(10 + i, 20 + i, 5.0 + i) for i in range(15)]
    eng = ScriptedEngine(faces, pups)
    names = pipeline.EYE_CASCADES + pipeline.MOUTH_CASCADES
    fd = pipeline.FaceDetector(cascade=b"c", puploc=b"pp", flploc={n: b"x" * (i + 1) for i, n in enumerate(names)}, engine=eng,
                               state=pipeline.DetectorState(lambda n: np.zeros(n, np.float32)))
    recs = fd.draw_faces(fd.detect_faces(np.zeros((300, 300), np.uint8)))
    assert len(recs) == 1 and len(recs[0].EyePoints) == 2 and len(recs[0].LandmarkPoints) == 15
    order = [(c[1], c[5]) for c in eng.calls if c[0] == "flp"]
    idx = {n: "plc:%d" % (i + 1) for i, n in enumerate(names)}
    want = [(idx[n], fl) for n in pipeline.EYE_CASCADES for fl in (False, True)] + [(idx[n], False) for n in pipeline.MOUTH_CASCADES] + [(idx["lp84"], True)]
    assert order == want  # cmd/pigo/main.go:492-563
    assert recs[0].LandmarkPoints[0] == pipeline.Coord(Row=20, Col=10, Scale=5)  # Col <- flp.Row, Row <- flp.Col
    with pytest.raises(core.PigoError, match="puploc cascade file is required"):
        pipeline.FaceDetector(cascade=b"c", flploc={"lp42": b"x"}, engine=ScriptedEngine(faces, [])).detect_faces(np.zeros((9, 9), np.uint8))


def _seeded_state(seed):
    counter = itertools.count()
    return pipeline.DetectorState(lambda n: synth.syn_uniform32(n, seed=seed, index=next(counter)))


def make_detector(engine, seed=1234, **kw):
    flp = {n: synth.cascade_bytes("lps/" + n) for n in pipeline.EYE_CASCADES + pipeline.MOUTH_CASCADES}
    return pipeline.FaceDetector(cascade=synth.facefinder_bytes(), puploc=synth.cascade_bytes("puploc"), flploc=flp, engine=engine,
                                 state=_seeded_state(seed), **kw)


def sample_rgba():
    g = synth.sample_gray()
    rgba = np.repeat(g[..., None], 4, axis=2)
    rgba[..., 3] = 255
    return rgba


def test_cli_flow_on_the_oracle_engine():
    """`pigo -in sample -cf facefinder -plc puploc -flpc lps -json -` with the CLI's default flags: one face, two eyes,
    fifteen landmark points (the count core/flploc_test.go:150-153 expects), coordinates inside the face box."""
    doc = make_detector(OracleEngine()).detect_json(sample_rgba())
    recs = json.loads(doc)
    assert doc.endswith("]\n") and " " not in doc and len(recs) == 1
    face = recs[0]["face"]
    assert list(recs[0].keys()) == ["eyes", "landmark_points", "face"]  # struct field order
    assert len(recs[0]["eyes"]) == 2 and len(recs[0]["landmark_points"]) == 15
    for pt in recs[0]["eyes"] + recs[0]["landmark_points"]:
        assert face["x"] <= pt["x"] <= face["x"] + face["size"] and face["y"] <= pt["y"] <= face["y"] + face["size"] + 20
    assert 200 < face["size"] < 300


def test_find_faces_on_the_oracle_engine():
    frame = synth.syn_faces(480, 640, seed=1234, frame_index=0)
    got = pipeline.find_faces(frame, engine=OracleEngine())
    assert got.dtype == np.int64 and got.ndim == 2 and got.shape[1] == 3 and len(got) >= 1
    assert (got[:, 2] >= 100).all() and (got[:, 2] <= 600).all()


def test_bench_verification_covers_both_ends_of_a_batch():
    """bench.py checks the first and the last frames of the timed batch against the oracle (the head of the XCD dealing and the
    `nframes % 8` remainder / last pipeline chunk)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.verify_frame_indices(128, 8) == [0, 1, 2, 3, 124, 125, 126, 127]
    assert bench.verify_frame_indices(13, 8) == [0, 1, 2, 3, 9, 10, 11, 12]
    assert bench.verify_frame_indices(5, 8) == [0, 1, 2, 3, 4]
    assert bench.verify_frame_indices(1, 8) == [0]
    assert bench.verify_frame_indices(1024, 4) == [0, 1, 1022, 1023]


def test_synthetic_frames_with_rotated_faces_are_seeded_and_differ_from_upright():
    """synth.make_frames(rotate_deg=...) -- the frames bench.py --face-rotation scans: deterministic, and not the upright ones."""
    from pigo_amd import synth
    a = synth.make_frames("faces", 2, 240, 320, seed=3, rotate_deg=-79.0)
    b = synth.make_frames("faces", 2, 240, 320, seed=3, rotate_deg=-79.0)
    c = synth.make_frames("faces", 2, 240, 320, seed=3)
    assert a.shape == (2, 240, 320) and a.dtype == c.dtype and (a == b).all() and not (a == c).all()
