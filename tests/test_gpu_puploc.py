"""GPU parity tests for pupil / facial-landmark localisation (core/puploc.go:106-277, core/flploc.go:36-57; SURVEY.md
section 8 rows f2/f3): the HIP kernel through the C ABI against the CPU oracle and the committed known answers.
Integer outputs bit-exact; the float32 scale is reproduced operation by operation, so it is compared bit-for-bit too
(contract tolerance 1e-5)."""
import hashlib

import numpy as np
import pytest

import oracle
from pigo_amd import core, synth

import landmark_pipeline as lp
from test_puploc_cpu import f32, iter_golden_cases, puploc_golden

pytestmark = pytest.mark.gpu


def _img(gray, rows, cols, dim=None):
    return core.ImageParams(Pixels=gray, Rows=rows, Cols=cols, Dim=dim or cols)


def test_unpack_header_matches_oracle():
    for name in ("puploc", "lps/lp42", "lps/lp93"):
        pk = synth.cascade_bytes(name)
        st, sc, tr, d = core.NewPuplocCascade(0).UnpackCascade(pk).header
        ost, osc, otr, od = oracle.OraclePuploc.unpack(pk).header
        assert (st, tr, d) == (ost, otr, od) and np.float32(sc) == np.float32(osc)


def test_puploc_goldens_through_the_c_abi():
    for case, casc, img, rnd, pool in iter_golden_cases(lambda n: core.NewPuplocCascade(0).UnpackCascade(synth.cascade_bytes(n))):
        rows, cols = img.shape
        got = casc.RunDetector(core.Puploc(case["row"], case["col"], case["scale"], case["perturbs"]), _img(img, rows, cols), case["angle"],
                               case["flip_v"], rnd=rnd, pool=pool)
        assert [got.Row, got.Col] == case["want"][:2] and np.float32(got.Scale) == f32(case["want"][2]), (case, got)
        assert got.Perturbs == 0
        assert hashlib.sha256(pool.tobytes()).hexdigest() == case["pool_after_sha"], case  # the sorted pool comes back bit-exact


def test_reference_landmark_sequence_matches_oracle_and_invariant():
    """core/flploc_test.go:84-153 end to end on the GPU: eyes + 15 landmark points, pool object threaded through."""
    gray = synth.sample_gray()
    g = puploc_golden()["sequence"]
    face = tuple(g["face"])
    hb, ob = lp.HipBackend(gray, 400, 320, 320), lp.OracleBackend(gray, 400, 320, 320)
    for seed in (g["seed"], 7):
        hs, os_ = lp.run_sequence(hb, face, seed=seed), lp.run_sequence(ob, face, seed=seed)
        assert hs["left"] == os_["left"] and hs["right"] == os_["right"]
        for x, y in zip(hs["points"], os_["points"]):
            assert x == y, (x, y)
        assert sum(1 for _, _, p in hs["points"] if p[0] > 0 and p[1] > 0) == 15
    hs = lp.run_sequence(hb, face, seed=g["seed"])
    for (n, fl, p), w in zip(hs["points"], g["points"]):
        assert [n, fl, p[0], p[1]] == w[:4] and p[2] == f32(w[4])


def test_random_requests_upright_rotated_flipped_borders():
    gray = synth.sample_gray()
    land = np.ascontiguousarray(gray[40:340, :])  # rows < cols would need a landscape crop: 300 x 320
    rng = np.random.default_rng(2024)
    for name in ("puploc", "lps/lp44"):
        pk = synth.cascade_bytes(name)
        h, o = core.NewPuplocCascade(0).UnpackCascade(pk), oracle.OraclePuploc.unpack(pk)
        ph, po = core.new_pool(), np.zeros((3, 63), np.float32)
        for t in range(40):
            img, rows, cols = (gray, 400, 320) if t % 3 else (land, 300, 320)
            P = int(rng.choice([63, 63, 50, 33, 1, 0]))
            row, col = int(rng.integers(-30, rows + 30)), int(rng.integers(-30, cols + 30))
            sc = float(np.float32(rng.uniform(1.0, 200.0)))
            ang = float(rng.choice([0.0, 0.0, 0.02, 0.25, 0.5, 0.77, 1.0, 3.0]))
            flip = bool(rng.integers(0, 2))
            rnd = synth.syn_uniform32(189, seed=31, index=t)
            use_pool = t % 4 != 0
            a = h.RunDetector(core.Puploc(row, col, sc, P), _img(img, rows, cols), ang, flip, rnd=rnd, pool=ph if use_pool else None)
            b = o.run_detector(row, col, sc, P, img, rows, cols, cols, ang, flip, rnd, po if use_pool else None)
            assert (a.Row, a.Col) == b[:2] and np.float32(a.Scale) == b[2], (name, t, a, b)
            assert abs(float(a.Scale) - float(b[2])) <= 1e-5
            assert (ph == po).all()


def test_dim_wider_than_cols_and_errors():
    gray = synth.sample_gray()
    wide = np.zeros((400, 352), np.uint8)
    wide[:, :320] = gray
    pk = synth.cascade_bytes("puploc")
    h, o = core.NewPuplocCascade(0).UnpackCascade(pk), oracle.OraclePuploc.unpack(pk)
    rnd = synth.syn_uniform32(189, seed=3)
    a = h.RunDetector(core.Puploc(187, 109, 65.25, 63), _img(wide, 400, 320, 352), 0.0, False, rnd=rnd)
    b = o.run_detector(187, 109, 65.25, 63, wide, 400, 320, 352, 0.0, False, rnd)
    assert (a.Row, a.Col, np.float32(a.Scale)) == b
    with pytest.raises(core.PigoPanic):  # det.rows[63] = res[0]: index out of range (puploc.go:262)
        h.RunDetector(core.Puploc(187, 109, 65.25, 64), _img(gray, 400, 320), 0.0, False, rnd=synth.syn_uniform32(192))
    with pytest.raises(core.PigoPanic):
        h.RunDetector(core.Puploc(187, 109, 65.25, -2), _img(gray, 400, 320), 0.0, False, rnd=rnd)
    with pytest.raises(core.PigoPanic):  # pixels shorter than the image
        h.RunDetector(core.Puploc(187, 109, 65.25, 63), _img(gray[:100], 400, 320), 0.0, False, rnd=rnd)
    with pytest.raises(ValueError):
        h.RunDetector(core.Puploc(187, 109, 65.25, 63), _img(gray, 400, 320, 300), 0.0, False, rnd=rnd)
    assert isinstance(h.RunDetector(core.Puploc(187, 109, 65.25, 63), _img(gray, 400, 320), 0.0, False), core.Puploc)  # rnd drawn internally


def test_batch_requests_over_device_frames():
    import torch
    from pigo_amd import batch
    rows, cols, nf = 240, 320, 6
    frames = synth.make_frames("faces", nf, rows, cols, seed=1234)
    d_frames = torch.from_numpy(frames).cuda()
    pk = synth.cascade_bytes("puploc")
    h, o = core.NewPuplocCascade(0).UnpackCascade(pk), oracle.OraclePuploc.unpack(pk)
    rng = np.random.default_rng(11)
    n = 300
    reqs = np.zeros(n, dtype=core.PUPLOC_REQ_DTYPE)
    reqs["row"], reqs["col"] = rng.integers(0, rows, n), rng.integers(0, cols, n)
    reqs["scale"] = rng.uniform(4, 60, n).astype(np.float32)
    reqs["perturbs"] = rng.choice([63, 50, 17], n)
    reqs["frame"], reqs["flip_v"] = rng.integers(0, nf, n), rng.integers(0, 2, n)
    rnd = np.stack([synth.syn_uniform32(189, seed=77, index=i) for i in range(n)])
    d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(n, 24)).cuda()
    d_rnd = torch.from_numpy(rnd).cuda()
    for angle in (0.0, 0.4):
        pool = torch.zeros((n, 189), dtype=torch.float32, device="cuda")
        out = batch.puploc_run_batch(h, d_frames, d_reqs, d_rnd, angle=angle, pool=pool)
        torch.cuda.synchronize()
        batch.puploc_status(h)
        got = out.cpu().numpy().view(core.PUPLOC_DTYPE).reshape(n)
        ph = pool.cpu().numpy()
        for i in range(n):
            r = reqs[i]
            po = np.zeros((3, 63), np.float32)
            want = o.run_detector(int(r["row"]), int(r["col"]), float(r["scale"]), int(r["perturbs"]), frames[r["frame"]], rows, cols, cols, angle,
                                  bool(r["flip_v"]), rnd[i], po)
            assert (int(got[i]["row"]), int(got[i]["col"])) == want[:2] and got[i]["scale"] == want[2], (angle, i, got[i], want)
            assert (ph[i] == po.ravel()).all()
    # without a pool tensor every request gets a brand-new pool object
    out = batch.puploc_run_batch(h, d_frames, d_reqs, d_rnd, angle=0.0)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(core.PUPLOC_DTYPE).reshape(n)
    r = reqs[5]
    want = o.run_detector(int(r["row"]), int(r["col"]), float(r["scale"]), int(r["perturbs"]), frames[r["frame"]], rows, cols, cols, 0.0,
                          bool(r["flip_v"]), rnd[5], None)
    assert (int(got[5]["row"]), int(got[5]["col"]), got[5]["scale"]) == want
    # a request the reference would panic on is reported, not silently computed
    bad = reqs[:4].copy()
    bad["perturbs"][2] = 64
    batch.puploc_run_batch(h, d_frames, torch.from_numpy(bad.view(np.uint8).reshape(4, 24)).cuda(), d_rnd[:4].contiguous())
    torch.cuda.synchronize()
    with pytest.raises(core.PigoPanic):
        batch.puploc_status(h)
    batch.puploc_status(h)  # the flag is cleared by reading it


def test_batch_requests_on_padded_frames_clamp_with_cols_not_dim():
    """Frames with Dim > Cols (the output of rgb_to_grayscale(dim=...)): the column clamp min(ncols-1, ...) of
    puploc.go:118-128 must use ImageParams.Cols, not the stride -- requests near the right border would otherwise read
    the padding bytes.  ADVICE r1: puploc_run_batch(cols=...)."""
    import torch
    from pigo_amd import batch
    rows, cols, dim, nf = 200, 300, 352, 3
    frames = np.full((nf, rows, dim), 255, dtype=np.uint8)  # bright padding: a clamp that lands in it changes the comparisons
    frames[:, :, :cols] = synth.make_frames("faces", nf, rows, cols, seed=5)
    d_frames = torch.from_numpy(frames).cuda()
    pk = synth.cascade_bytes("puploc")
    h, o = core.NewPuplocCascade(0).UnpackCascade(pk), oracle.OraclePuploc.unpack(pk)
    rng = np.random.default_rng(3)
    n = 120
    reqs = np.zeros(n, dtype=core.PUPLOC_REQ_DTYPE)
    reqs["row"], reqs["col"] = rng.integers(0, rows, n), rng.integers(cols - 40, cols + 10, n)  # hugging the right border
    reqs["scale"] = rng.uniform(20, 90, n).astype(np.float32)
    reqs["perturbs"] = 63
    reqs["frame"], reqs["flip_v"] = rng.integers(0, nf, n), rng.integers(0, 2, n)
    rnd = np.stack([synth.syn_uniform32(189, seed=9, index=i) for i in range(n)])
    d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(n, 24)).cuda()
    d_rnd = torch.from_numpy(rnd).cuda()
    differs = 0
    for angle in (0.0, 0.3):
        out = batch.puploc_run_batch(h, d_frames, d_reqs, d_rnd, angle=angle, cols=cols)
        torch.cuda.synchronize()
        batch.puploc_status(h)
        got = out.cpu().numpy().view(core.PUPLOC_DTYPE).reshape(n)
        for i in range(n):
            r = reqs[i]
            want = o.run_detector(int(r["row"]), int(r["col"]), float(r["scale"]), 63, frames[r["frame"]], rows, cols, dim, angle,
                                  bool(r["flip_v"]), rnd[i], None)
            assert (int(got[i]["row"]), int(got[i]["col"])) == want[:2] and got[i]["scale"] == want[2], (angle, i, got[i], want)
            wrong = o.run_detector(int(r["row"]), int(r["col"]), float(r["scale"]), 63, frames[r["frame"]], rows, dim, dim, angle,
                                   bool(r["flip_v"]), rnd[i], None)
            differs += wrong != want
    assert differs > 0  # the test would not notice the bug otherwise
