"""GPU tests of the result plumbing (scope row f4): the CLI flow and FindFaces on the HIP engine must produce the same
documents as on the oracle-backed engine, byte for byte."""
import json

import numpy as np
import pytest

from pigo_amd import core, pipeline, synth

from oracle_engine import OracleEngine
from test_pipeline_cpu import make_detector, sample_rgba

pytestmark = pytest.mark.gpu


def test_cli_json_document_matches_oracle_engine():
    rgba = sample_rgba()
    for seed, kw in ((1234, {}), (5, dict(shiftFactor=0.1, scaleFactor=1.1, iouThreshold=0.1)), (9, dict(angle=0.0, minSize=40))):
        got = make_detector(pipeline.HipEngine(0), seed=seed, **kw).detect_json(rgba)
        want = make_detector(OracleEngine(), seed=seed, **kw).detect_json(rgba)
        assert got == want and got
        rec = json.loads(got)[0]
        assert len(rec["eyes"]) == 2 and len(rec["landmark_points"]) == 15


def test_cli_json_many_faces_cumulative_lists():
    """A frame with several faces: the eyes / landmark lists grow from face to face (cmd/pigo/main.go:363-366)."""
    frame = synth.syn_faces(480, 640, seed=1234, frame_index=0)
    kw = dict(minSize=40, maxSize=400, shiftFactor=0.1, scaleFactor=1.1, iouThreshold=0.2)
    got = make_detector(pipeline.HipEngine(0), seed=3, **kw).detect_json(frame)
    want = make_detector(OracleEngine(), seed=3, **kw).detect_json(frame)
    assert got == want
    recs = json.loads(got)
    assert len(recs) >= 2
    sizes = [len(r.get("eyes", [])) for r in recs]
    assert sizes == sorted(sizes) and sizes[-1] >= 2


def test_find_faces_matches_oracle_engine():
    for fi in (0, 1):
        frame = synth.syn_faces(480, 640, seed=1234, frame_index=fi)
        got = pipeline.find_faces(frame)
        want = pipeline.find_faces(frame, engine=OracleEngine())
        assert got.shape == want.shape and (got == want).all()
