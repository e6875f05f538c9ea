"""CPU tests for scope rows f2/f3 (pupil / facial-landmark localisation, core/puploc.go + core/flploc.go): the oracle
against the committed known answers, the NumPy restatement and the reference's own test invariants; and the C ABI's
argument checks that run before any device call."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import oracle
from oracle.np_restatement import NpPuploc
from pigo_amd import core, synth

import landmark_pipeline as lp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def puploc_golden():
    with open(os.path.join(ROOT, "tests", "golden", "puploc_golden.json")) as fh:
        return json.load(fh)


def golden_img(name):
    return synth.sample_gray() if name == "sample" else synth.syn_noise(240, 320, seed=1234, frame_index=3)


def f32(hexs):
    return np.frombuffer(bytes.fromhex(hexs), dtype="<f4")[0]


def iter_golden_cases(make_cascade):
    """Yields (case, cascade, img, rnd, pool) with the 'chain' pool threading of make_puploc_golden.py."""
    cascs, pool = {}, None
    for case in puploc_golden()["cases"]:
        if case["cascade"] not in cascs:
            cascs[case["cascade"]] = make_cascade(case["cascade"])
        if case["pool"] == "fresh" or pool is None:
            pool = np.zeros((3, 63), np.float32)
        yield case, cascs[case["cascade"]], golden_img(case["image"]), synth.syn_uniform32(3 * 63, seed=1234, index=case["rnd_index"]), pool


def test_unpack_wire_format_f3():
    """Row f3: {stages u32, scale f32, trees u32, depth u32} then per tree 4*2^d-4 code bytes + 2*2^d float32 (puploc.go:38-103)."""
    for name, want in (("puploc", (5, 20, 10)), ("lps/lp42", (6, 20, 9)), ("lps/lp84", (6, 20, 9))):
        pk = synth.cascade_bytes(name)
        o = oracle.OraclePuploc.unpack(pk)
        st, sc, tr, d = o.header
        assert (st, tr, d) == want and (st, tr, d) == struct.unpack("<I4xII", pk[:16])
        assert np.float32(sc) == struct.unpack("<f", pk[4:8])[0]
        D = 1 << d
        assert len(pk) == 16 + st * tr * (12 * D - 4)
        codes, preds = o.tables()
        n = NpPuploc(pk)
        assert (codes == n.codes.ravel()).all() and (preds == n.preds.ravel()).all()
        # first tree, first node and first leaf pair straight from the file
        assert (codes[:4] == np.frombuffer(pk[16:20], dtype=np.int8)).all()
        assert (preds[:2] == np.frombuffer(pk[16 + 4 * D - 4: 16 + 4 * D + 4], dtype="<f4")).all()
    for cut in (0, 15, 16, 5000, len(pk) - 1):
        with pytest.raises(oracle.OraclePanic):
            oracle.OraclePuploc.unpack(pk[:cut])


def test_oracle_against_puploc_goldens():
    for case, casc, img, rnd, pool in iter_golden_cases(lambda n: oracle.OraclePuploc.unpack(synth.cascade_bytes(n))):
        rows, cols = img.shape
        got = casc.run_detector(case["row"], case["col"], case["scale"], case["perturbs"], img, rows, cols, cols, case["angle"], case["flip_v"],
                                rnd, pool)
        assert [got[0], got[1]] == case["want"][:2] and got[2] == f32(case["want"][2]), case
        assert hashlib.sha256(pool.tobytes()).hexdigest() == case["pool_after_sha"]


def test_oracle_vs_numpy_random_requests():
    gray = synth.sample_gray()
    rng = np.random.default_rng(77)
    for name in ("puploc", "lps/lp312"):
        pk = synth.cascade_bytes(name)
        o, n = oracle.OraclePuploc.unpack(pk), NpPuploc(pk)
        po, pn = np.zeros((3, 63), np.float32), np.zeros((3, 63), np.float32)
        for t in range(12):
            P = int(rng.choice([63, 63, 50, 31, 2]))
            row, col, sc = int(rng.integers(-20, 420)), int(rng.integers(-20, 340)), float(np.float32(rng.uniform(2, 150)))
            ang = float(rng.choice([0.0, 0.0, 0.25, 0.9, 1.0]))
            flip = bool(t & 1)
            rnd = synth.syn_uniform32(189, seed=9, index=t)
            a = o.run_detector(row, col, sc, P, gray, 400, 320, 320, ang, flip, rnd, po)
            b = n.run_detector(row, col, sc, P, gray, 400, 320, 320, ang, flip, rnd, pn)
            assert a == b and (po == pn).all(), (name, t, a, b)


def test_pool_quirk_first_call_is_not_the_median():
    """puploc.go:267-275 sorts all 63 pool entries: with Perturbs=50 on a new pool object the 13 untouched zeros sort to
    the front and index 25 is the 12th smallest result, not the median; on the next call the stale entries are the 13
    largest results of the previous call, so index 25 is the 25th smallest of the new results."""
    gray = synth.sample_gray()
    o = oracle.OraclePuploc.unpack(synth.cascade_bytes("puploc"))
    rnd = synth.syn_uniform32(189, seed=5, index=0)
    rr = np.array([o.classify(*(np.float32(v) for v in (
        np.float32(187) + (np.float32(65.25) * np.float32(0.15)) * (np.float32(0.5) - rnd[3 * i]),
        np.float32(109) + (np.float32(65.25) * np.float32(0.15)) * (np.float32(0.5) - rnd[3 * i + 1]),
        np.float32(65.25) * (np.float32(0.925) + np.float32(0.15) * rnd[3 * i + 2]))), gray, 400, 320, 320) for i in range(50)])
    rows_sorted = np.sort(rr[:, 0])
    pool = np.zeros((3, 63), np.float32)
    first = o.run_detector(187, 109, 65.25, 50, gray, 400, 320, 320, 0.0, False, rnd, pool)
    assert first[0] == int(rows_sorted[25 - 13])
    second = o.run_detector(187, 109, 65.25, 50, gray, 400, 320, 320, 0.0, False, rnd, pool)
    assert second[0] == int(rows_sorted[25])
    with pytest.raises(oracle.OraclePanic):  # det.rows[63] = ...: index out of range
        o.run_detector(187, 109, 65.25, 64, gray, 400, 320, 320, 0.0, False, synth.syn_uniform32(3 * 64), None)


def test_reference_landmark_invariant_any_seed():
    """core/flploc_test.go:84-153: 2*5 + 4 + 1 = 15 landmark points with Row > 0 and Col > 0 on the sample face;
    core/puploc_test.go:34-80: eyes are found.  Must hold whatever math/rand hands out."""
    gray = synth.sample_gray()
    g = puploc_golden()["sequence"]
    face = tuple(g["face"])
    be = lp.OracleBackend(gray, 400, 320, 320)
    for seed in (1234, 1, 99):
        seq = lp.run_sequence(be, face, seed=seed)
        assert sum(1 for _, _, p in seq["points"] if p[0] > 0 and p[1] > 0) == 15
        assert 150 < seq["left"][0] < 215 and 80 < seq["left"][1] < 150 and 160 < seq["right"][1] < 230  # eyes inside the face box
    seq = lp.run_sequence(be, face, seed=g["seed"])
    assert [seq["left"][0], seq["left"][1]] == g["left"][:2] and seq["left"][2] == f32(g["left"][2])
    assert [seq["right"][0], seq["right"][1]] == g["right"][:2] and seq["right"][2] == f32(g["right"][2])
    for (n, fl, p), w in zip(seq["points"], g["points"]):
        assert [n, fl, p[0], p[1]] == w[:4] and p[2] == f32(w[4])


def test_puploc_abi_argument_checks_need_no_gpu():
    pk = synth.cascade_bytes("puploc")
    plc = core.NewPuplocCascade(0)
    for cut in (0, 15, 16, 4000, len(pk) - 1):  # the reference panics on a short packet (puploc.go:51-88)
        with pytest.raises(core.PigoPanic):
            plc.UnpackCascade(pk[:cut])
    bad = bytearray(pk)
    bad[12:16] = (40).to_bytes(4, "little")
    with pytest.raises(ValueError):
        plc.UnpackCascade(bytes(bad))
    L = core.load_library()
    if L.pigo_device_count() == 0:  # no GPU: a good packet must fail with a HIP error, never fall back
        with pytest.raises(core.PigoError):
            plc.UnpackCascade(pk)
    with pytest.raises(core.PigoError):
        plc.RunDetector(core.Puploc(1, 1, 10.0, 63), core.ImageParams(np.zeros(100, np.uint8), 10, 10, 10), 0.0, False)  # not unpacked
    assert core.PUPLOC_DTYPE.itemsize == 16 and core.PUPLOC_REQ_DTYPE.itemsize == 24
    assert core.draw_perturbations(63).dtype == np.float32 and core.draw_perturbations(63).size == 189
    assert core.new_pool().shape == (3, 63)
