"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed goldens.

Bar: integer (row, col, scale) bit-exact, in the reference's order; |dq| <= 1e-5 -- the
implementation targets 0 ulp, so raw-detection q is compared bit-for-bit (Q_TOL_RAW = 0.0) and cluster q,
a float32 sum reproduced in the reference's order, likewise; the 1e-5 contract tolerance is asserted
separately so a future relaxation of the design target does not silently weaken the contract.
"""
import os

import numpy as np
import pytest

import oracle
from pigo_amd import core, synth

from conftest import assert_same_dets, f32_from_hex, golden_image

pytestmark = pytest.mark.gpu

Q_TOL_CONTRACT = 1e-5  # north_star: "within 1e-5 on the float q score"
Q_TOL_RAW = 0.0        # design target: bit-exact

# scan implementations of the release library: 3 = LDS regions (default where eligible), 2 = LDS tiles, 0 = monolithic
# fallback.  Variant 1 (the first head + tail design) is compiled into the debug build only (python -m pigo_amd.build --debug;
# PIGO_HIP_LIB selects it).
VARIANTS = [3, 2, 0] + ([1] if "debug" in core.library_path() else [])


def _cp(img, rows, cols, dim, mn, mx, shift, scale):
    return core.CascadeParams(MinSize=mn, MaxSize=mx, ShiftFactor=shift, ScaleFactor=scale,
                              ImageParams=core.ImageParams(Pixels=img, Rows=rows, Cols=cols, Dim=dim))


def _golden_dets(rows):
    return core.make_dets([(r, c, s, f32_from_hex(q)) for r, c, s, q in rows])


def _as_core(d):
    return core.make_dets([(int(a["row"]), int(a["col"]), int(a["scale"]), a["q"]) for a in d])


def _as_oracle(d):
    return oracle.make_dets([(int(a["row"]), int(a["col"]), int(a["scale"]), a["q"]) for a in d])


# ---- Unpack -------------------------------------------------------------------------------------------------


def test_unpack_tables_match_oracle(pg, orc):
    assert pg.treeDepth == orc.tree_depth == 6 and pg.treeNum == orc.tree_num == 468
    for a, b in zip(pg.tables(), orc.tables()):
        assert a.shape == b.shape and (a == b).all()


# ---- RunCascade + ClusterDetections against the goldens (both scan variants) -----------------------------------


@pytest.mark.parametrize("variant", VARIANTS)
def test_golden_cases(pg, golden, variant, monkeypatch):
    monkeypatch.setenv("PIGO_SCAN_VARIANT", str(variant))
    fresh = core.NewPigo(0).Unpack(synth.facefinder_bytes())  # new handle: plans are cached per handle
    for case in golden["cases"]:
        img = golden_image(case["name"])
        cp = _cp(img, case["rows"], case["cols"], case["dim"], case["min_size"], case["max_size"], case["shift"], case["scale"])
        dets = fresh.RunCascade(cp, case["angle"])
        assert_same_dets(dets, _golden_dets(case["detections"]), f"{case['name']} v{variant}", Q_TOL_RAW)
        assert_same_dets(dets, _golden_dets(case["detections"]), f"{case['name']} v{variant}", Q_TOL_CONTRACT)
        cl = fresh.ClusterDetections(dets, case["iou"])
        assert_same_dets(cl, _golden_dets(case["clusters"]), f"{case['name']} clusters", Q_TOL_RAW)


def test_reference_test_invariants(pg, gray):
    """core/pigo_test.go:68-84 and core/flploc_test.go:102-153, run through the product."""
    dets = pg.RunCascade(_cp(gray, 400, 320, 320, 20, 1000, 0.2, 1.1), 0.0)
    cl = pg.ClusterDetections(dets, 0.1)
    assert len(cl) > 0 and int((cl["scale"] > 50).sum()) == 1


# ---- seeded sweeps against the oracle -------------------------------------------------------------------------


@pytest.mark.parametrize("variant", VARIANTS)
def test_random_parameter_sweep(orc, variant, monkeypatch):
    monkeypatch.setenv("PIGO_SCAN_VARIANT", str(variant))
    fresh = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    rng = np.random.default_rng(2024 + variant)
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
        want = orc.run_cascade(img, rows, cols, dim, mn, mx, shift, scale, angle)
        got = fresh.RunCascade(_cp(img, rows, cols, dim, mn, mx, shift, scale), angle)
        assert_same_dets(got, want, f"sweep {k}: {rows}x{cols} dim {dim} [{mn},{mx}] {shift}/{scale} a={angle} v{variant}", Q_TOL_RAW)
        iou = float(rng.choice([0.0, 0.01, 0.1, 0.15, 0.2]))  # thresholds used by the reference's callers (Appendix E)
        wc, ties = orc.cluster_detections(want.copy(), iou, want_ties=True)
        gc = fresh.ClusterDetections(got, iou)
        assert_same_dets(gc, wc, f"sweep {k} clusters (ties={ties})", Q_TOL_RAW)


def test_edge_cases(pg, orc):
    # image smaller than the smallest window: the ladder is empty -> no detections, no error (pigo.go:230)
    tiny = synth.syn_noise(12, 12, seed=1)
    assert len(pg.RunCascade(_cp(tiny, 12, 12, 12, 20, 1000, 0.1, 1.1), 0.0)) == 0
    # MaxSize < MinSize: the loop never runs
    img = synth.syn_faces(120, 160, seed=3)
    assert len(pg.RunCascade(_cp(img, 120, 160, 160, 50, 20, 0.1, 1.1), 0.0)) == 0
    # a single scale, a single row / column of windows
    for rows, cols, s in ((41, 200, 38), (200, 41, 38), (41, 41, 38)):
        im = synth.syn_faces(rows, cols, seed=5)
        want = orc.run_cascade(im, rows, cols, cols, s, s, 0.1, 1.1, 0.0)
        assert_same_dets(pg.RunCascade(_cp(im, rows, cols, cols, s, s, 0.1, 1.1), 0.0), want, f"single {rows}x{cols}")
    # scale factor <= 1 still terminates (grows by 2, pigo.go:255), shift tiny -> step 1
    im = synth.syn_faces(90, 90, seed=6)
    want = orc.run_cascade(im, 90, 90, 90, 20, 60, 0.001, 0.5, 0.0)
    assert_same_dets(pg.RunCascade(_cp(im, 90, 90, 90, 20, 60, 0.001, 0.5), 0.0), want, "scale<1")
    # flat image: every window dies at tree 0; all-255 / all-0
    for v in (0, 255):
        flat = np.full((100, 100), v, dtype=np.uint8)
        assert len(pg.RunCascade(_cp(flat, 100, 100, 100, 20, 1000, 0.1, 1.1), 0.0)) == 0
    # parameter errors instead of undefined behaviour
    with pytest.raises(ValueError):
        pg.RunCascade(_cp(im, 90, 90, 80, 20, 60, 0.1, 1.1), 0.0)  # dim < cols
    with pytest.raises(ValueError):
        pg.RunCascade(_cp(im[:40], 90, 90, 90, 20, 60, 0.1, 1.1), 0.0)  # len(pixels) < rows*dim
    with pytest.raises(ValueError):
        pg.RunCascade(_cp(im, 90, 90, 90, 20, 60, float("nan"), 1.1), 0.0)


def test_rotated_portrait_reads_like_go(pg, orc, gray):
    """Quirk Q1 on a portrait frame: columns clamp at nrows-1, i.e. the scan reads into the next row -- the
    guarded kernel must give the oracle's answer and must not raise."""
    for a in (0.1, 0.8, 1.0):
        want = orc.run_cascade(gray, 400, 320, 320, 20, 1000, 0.1, 1.1, a)
        assert_same_dets(pg.RunCascade(_cp(gray, 400, 320, 320, 20, 1000, 0.1, 1.1), a), want, f"portrait a={a}", Q_TOL_RAW)


def test_cluster_semantics_and_ties(pg, orc):
    # hand-made lists, including the quirk-Q5 chain and exact Q ties (order then depends on Go's pdqsort,
    # which both sides restate; n <= 12 is the stable insertion-sort regime)
    lists = [
        [],
        [(100, 100, 50, 1.0)],
        [(100, 100, 50, 1.0), (104, 100, 50, 2.0), (108, 100, 50, 3.0), (300, 300, 40, 4.0)],
        [(100, 100, 100, 1.0), (100, 140, 100, 2.0), (100, 180, 100, 3.0)],
        [(10, 10, 20, 2.5), (11, 10, 20, 2.5), (200, 200, 20, 2.5), (201, 200, 20, 2.5), (12, 10, 20, 1.0)],
    ]
    for rows in lists:
        for iou in (0.0, 0.01, 0.1, 0.2, 0.5, 1.0):
            a, b = core.make_dets(rows), oracle.make_dets(rows)
            got, want = pg.ClusterDetections(a, iou), orc.cluster_detections(b, iou)
            assert_same_dets(got, want, f"clusters of {rows} @ {iou}", Q_TOL_RAW)
            assert_same_dets(a, b, "in-place sorted input", Q_TOL_RAW)  # the reference sorts the caller's slice
    # large tie-heavy list: thousands of detections on a grid with few distinct Q values
    rng = np.random.default_rng(9)
    n = 3000
    rows = [(int(rng.integers(50, 1000)), int(rng.integers(50, 1800)), int(rng.choice([40, 60, 90, 140])),
             float(rng.integers(1, 40)) / 4.0) for _ in range(n)]
    a, b = core.make_dets(rows), oracle.make_dets(rows)
    got, want = pg.ClusterDetections(a, 0.2), orc.cluster_detections(b, 0.2)
    assert_same_dets(a, b, "sorted 3000", Q_TOL_RAW)
    assert_same_dets(got, want, "clusters of 3000", Q_TOL_RAW)


@pytest.mark.parametrize("v2", ["0", "1"])
def test_cluster_kernels_agree_on_sweep_lists(pg, orc, v2, monkeypatch):
    """Both ClusterDetections implementations -- k_cluster (one workgroup per frame) and the seeds / members / compact kernels
    for long lists -- against the oracle on the same lists: random boxes at every threshold the reference's callers use
     plus 1.0 and a negative one, scale-0 windows (IoU with itself is NaN: a seed without a cluster),
    lists that straddle the 256-candidate block of k_cluster_seeds.  core/pigo.go:262-308."""
    monkeypatch.setenv("PIGO_CLUSTER_V2", v2)
    rng = np.random.default_rng(77)
    for n in (1, 2, 13, 64, 255, 256, 257, 700, 1500):
        rows = [(int(rng.integers(20, 400)), int(rng.integers(20, 600)), int(rng.choice([0, 20, 24, 40, 60, 90, 140])),
                 float(rng.integers(1, 4000)) / 16.0) for _ in range(n)]
        for iou in (0.0, 0.01, 0.15, 0.2, 0.6, 1.0, -0.5):
            a, b = core.make_dets(rows), oracle.make_dets(rows)
            got, want = pg.ClusterDetections(a, iou), orc.cluster_detections(b, iou)
            assert_same_dets(a, b, f"sorted n={n}", Q_TOL_RAW)
            assert_same_dets(got, want, f"clusters n={n} iou={iou} v2={v2}", Q_TOL_RAW)


def test_cluster_detections_has_no_length_limit(pg, orc):
    """The reference's ClusterDetections takes a slice of any length (core/pigo.go:262); round 2 refused more than 65,536
    detections.  70,000 boxes spread over a large canvas (so that the O(seeds x n) sweep stays small), ties included."""
    rng = np.random.default_rng(5)
    n = 70000
    rows = [(int(r), int(c), int(s), float(q) / 8.0) for r, c, s, q in
            zip(rng.integers(100, 60000, n), rng.integers(100, 60000, n), rng.choice([400, 640, 900, 1500], n), rng.integers(1, 2000, n))]
    a, b = core.make_dets(rows), oracle.make_dets(rows)
    got, want = pg.ClusterDetections(a, 0.2), orc.cluster_detections(b, 0.2)
    assert_same_dets(a, b, "sorted 70000", Q_TOL_RAW)
    assert len(want) > 100
    assert_same_dets(got, want, "clusters of 70000", Q_TOL_RAW)


# ---- full-size configs (BASELINE.json configs 2, 4, 5) ------------------------------------------------------------


def test_1080p_config2_and_4_against_oracle(pg, orc):
    for kind in ("faces", "noise"):
        f = synth.make_frames(kind, 1, 1080, 1920, seed=1234)[0]
        for angle in (0.0, 0.8):
            want = orc.run_cascade(f, 1080, 1920, 1920, 20, 1000, 0.1, 1.1, angle)
            got = pg.RunCascade(_cp(f, 1080, 1920, 1920, 20, 1000, 0.1, 1.1), angle)
            assert_same_dets(got, want, f"1080p {kind} angle={angle}", Q_TOL_RAW)
            wc = orc.cluster_detections(want.copy(), 0.2)
            assert_same_dets(pg.ClusterDetections(got, 0.2), wc, f"1080p {kind} clusters", Q_TOL_RAW)


def test_batch_api_matches_single_frame_and_is_order_stable(pg, orc):
    """pigo_plan_run on HBM-resident frames: every frame of a batch equals its own RunCascade result, for
    both scan variants, with a batch size that exercises the XCD frame dealing (13 = 8 + 5)."""
    import torch
    from pigo_amd import batch
    n, rows, cols = 13, 270, 480
    frames = synth.make_frames("faces", n, rows, cols, seed=77)
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(frames).to(dev)
    want = [orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, 0.0) for f in range(n)]
    for variant in VARIANTS:
        plan = batch.ScanPlan(pg, rows, cols, MinSize=20, MaxSize=1000, ShiftFactor=0.1, ScaleFactor=1.1, max_frames=n, det_cap=512)
        plan.set_variant(variant)
        dets, counts = plan.alloc_outputs(n)
        for rep in range(2):  # rerun on the same buffers: counters must be reset by the run itself
            plan.run(d_frames, dets, counts)
            torch.cuda.synchronize()
            plan.status()
            got = batch.dets_to_numpy(dets, counts)
            for f in range(n):
                assert_same_dets(got[f], want[f], f"batch frame {f} v{variant} rep{rep}", Q_TOL_RAW)
        inf = plan.info()
        assert inf.variant == variant and inf.windows_per_frame > 0
        # GPU-side per-frame clustering (stable sort; equal to Go's on tie-free lists)
        sorted_, clusters, ccounts, ties = plan.cluster(dets, counts, 0.2)
        torch.cuda.synchronize()
        cl = batch.dets_to_numpy(clusters, ccounts)
        srt = batch.dets_to_numpy(sorted_, counts)
        for f in range(n):
            w = want[f].copy()
            wc, wties = orc.cluster_detections(w, 0.2, want_ties=True)
            assert int(ties[f]) == wties
            assert_same_dets(cl[f], wc, f"batch clusters frame {f}", Q_TOL_RAW)   # tie frames are re-sorted with Go's pdqsort on the GPU
            assert_same_dets(srt[f], w, f"batch sorted frame {f}", Q_TOL_RAW)


def test_batch_clustering_is_tie_exact(pg, orc):
    """pigo_plan_cluster on lists with many tied Q values: the device re-sorts such frames with Go's (unstable) pdqsort, so
    the sorted lists, the seed order and the float32 sums equal the reference's -- for lists sorted out of LDS (<= 2048)
    and in global memory (> 2048), next to a tie-free and an empty frame."""
    import torch
    from pigo_amd import batch
    cap, nfr = 4096, 5
    plan = batch.ScanPlan(pg, 240, 320, MinSize=20, MaxSize=200, ShiftFactor=0.1, ScaleFactor=1.1, max_frames=nfr, det_cap=cap)
    rng = np.random.default_rng(4)
    lists = []
    for n, levels in ((317, 40), (1500, 9), (3000, 25), (200, 10**7), (0, 3)):
        rows = rng.integers(20, 220, n)
        cols = rng.integers(20, 300, n)
        scales = rng.integers(20, 90, n)
        q = (rng.integers(1, levels + 1, n) / np.float32(3.0)).astype(np.float32)
        lists.append(core.make_dets([(int(rows[i]), int(cols[i]), int(scales[i]), q[i]) for i in range(n)]))
    host = np.zeros((nfr, cap), dtype=core.DET_DTYPE)
    for f, l in enumerate(lists):
        host[f, : len(l)] = l
    dets = torch.from_numpy(host.view(np.int32).reshape(nfr, cap, 4)).cuda()
    counts = torch.tensor([len(l) for l in lists], dtype=torch.int32, device="cuda")
    sorted_, clusters, ccounts, ties = plan.cluster(dets, counts, 0.3)
    torch.cuda.synchronize()
    cl = batch.dets_to_numpy(clusters, ccounts)
    srt = batch.dets_to_numpy(sorted_, counts)
    for f, l in enumerate(lists):
        w = _as_oracle(l)
        wc, wties = orc.cluster_detections(w, 0.3, want_ties=True)
        assert int(ties[f]) == wties and (wties > 0) == (f < 3)
        assert_same_dets(srt[f], w, f"tie-exact sorted frame {f}", Q_TOL_RAW)
        assert_same_dets(cl[f], wc, f"tie-exact clusters frame {f}", Q_TOL_RAW)


def test_batch_go_order_sort_of_long_lists(pg, orc):
    """k_gosort_ties with one workgroup per frame (wave-parallel partitions, parts shared between the waves): lists of the 4K
    stress config's length and beyond -- out of LDS (<= 14336 keys) and out of the global workspace (> 14336) -- with ties of
    every density, already sorted / nearly sorted / descending inputs (partialInsertionSort, reverseRange, partitionEqual,
    breakPatterns), next to short lists.  Sorted lists and clusters must equal the oracle's restatement of sort.Slice +
    ClusterDetections (core/pigo.go:262-308)."""
    import torch
    from pigo_amd import batch
    cap = 20000
    rng = np.random.default_rng(11)
    specs = [(13000, 50, "rand"), (18000, 7, "rand"), (14336, 2000, "rand"), (14337, 10**7, "onetie"), (6000, 300, "asc"), (6000, 300, "nearly"),
             (5000, 40, "desc"), (9000, 1, "rand"), (64, 3, "rand"), (13, 2, "rand"), (12, 2, "rand"), (3000, 10**7, "rand")]
    nfr = len(specs)
    plan = batch.ScanPlan(pg, 240, 320, MinSize=20, MaxSize=200, ShiftFactor=0.1, ScaleFactor=1.1, max_frames=nfr, det_cap=cap)
    lists = []
    for n, levels, kind in specs:
        rows = rng.integers(20, 30000, n)
        cols = rng.integers(20, 30000, n)
        scales = rng.integers(20, 400, n)
        q = (rng.integers(1, levels + 1, n) / np.float32(3.0)).astype(np.float32)
        if kind == "onetie":
            q[rng.integers(0, n)] = q[rng.integers(0, n)]
        if kind in ("asc", "nearly"):
            q = np.sort(q)
        if kind == "nearly":
            for _ in range(4):
                i, j = rng.integers(0, n, 2)
                q[i], q[j] = q[j], q[i]
        if kind == "desc":
            q = np.sort(q)[::-1].copy()
        d = np.zeros(n, dtype=core.DET_DTYPE)
        d["row"], d["col"], d["scale"], d["q"] = rows, cols, scales, q
        lists.append(d)
    host = np.zeros((nfr, cap), dtype=core.DET_DTYPE)
    for f, l in enumerate(lists):
        host[f, : len(l)] = l
    dets = torch.from_numpy(host.view(np.int32).reshape(nfr, cap, 4)).cuda()
    counts = torch.tensor([len(l) for l in lists], dtype=torch.int32, device="cuda")
    for rep in range(2):
        sorted_, clusters, ccounts, ties = plan.cluster(dets, counts, 0.3)
        torch.cuda.synchronize()
        cl = batch.dets_to_numpy(clusters, ccounts)
        srt = batch.dets_to_numpy(sorted_, counts)
        for f, l in enumerate(lists):
            w = _as_oracle(l)
            wc, wties = orc.cluster_detections(w, 0.3, want_ties=True)
            assert int(ties[f]) == wties, (f, specs[f])
            assert_same_dets(srt[f], w, f"long sorted frame {f} {specs[f]}", Q_TOL_RAW)
            assert_same_dets(cl[f], wc, f"long clusters frame {f} {specs[f]}", Q_TOL_RAW)


@pytest.mark.parametrize("chunks,angle", [(2, 0.0), (3, 0.0), (1, 0.0), (0, 0.0), (2, 0.6)])
def test_chunked_pipeline_large_batches(pg, orc, chunks, angle, monkeypatch):
    """Batches of >= 16 frames are cut into chunks whose deep tail overlaps the next chunk's tile kernels (two queue sets,
    a tail stream).  29 frames = chunks of 16 + 13 (or 16 + 8 + 5): ragged last chunk, fewer than 8 frames in it, a second
    run on the same buffers, and the result must not depend on the chunk count."""
    import torch
    from pigo_amd import batch
    monkeypatch.setenv("PIGO_PIPE_CHUNKS", str(chunks))
    n, rows, cols = 29, 240, 320
    frames = np.concatenate([synth.make_frames("faces", n - 6, rows, cols, seed=5), synth.make_frames("noise", 6, rows, cols, seed=6)])
    d_frames = torch.from_numpy(frames).cuda()
    want = [orc.run_cascade(frames[f], rows, cols, cols, 20, 300, 0.1, 1.1, angle) for f in range(n)]
    plan = batch.ScanPlan(pg, rows, cols, MinSize=20, MaxSize=300, ShiftFactor=0.1, ScaleFactor=1.1, angle=angle, max_frames=32, det_cap=1024)
    dets, counts = plan.alloc_outputs(n)
    for rep in range(3):
        plan.run(d_frames, dets, counts)
        torch.cuda.synchronize()
        plan.status()
        got = batch.dets_to_numpy(dets, counts)
        for f in range(n):
            assert_same_dets(got[f], want[f], f"chunks={chunks} angle={angle} frame {f} rep{rep}", Q_TOL_RAW)
    assert sum(len(w) for w in want) > (50 if angle == 0.0 else 5)


@pytest.mark.parametrize("rules", ["5,16,20480", "6,32,40000;6,8,60000", "6,8,4096"])
def test_tile_geometry_rules(orc, rules, monkeypatch):
    """Variant 2 with unusual tile geometries (32-wide tiles, 2048-window tiles, almost everything on the
    global-memory class): the result may not depend on how the index space is tiled."""
    monkeypatch.setenv("PIGO_TILE_RULES", rules)
    monkeypatch.setenv("PIGO_SCAN_VARIANT", "2")
    fresh = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    for k, (rows, cols) in enumerate(((400, 320), (270, 480))):
        img = synth.sample_gray() if k == 0 else synth.syn_faces(rows, cols, seed=8, frame_index=k)
        want = orc.run_cascade(img, rows, cols, cols, 20, 1000, 0.1, 1.1, 0.0)
        got = fresh.RunCascade(_cp(img, rows, cols, cols, 20, 1000, 0.1, 1.1), 0.0)
        assert_same_dets(got, want, f"rules {rules} {rows}x{cols}", Q_TOL_RAW)
    # dim not a multiple of 4: the LDS path is not eligible, every rung takes the global-memory class
    img = np.zeros((200, 301), dtype=np.uint8)
    img[:, :300] = synth.syn_faces(200, 300, seed=9)
    want = orc.run_cascade(img, 200, 300, 301, 20, 1000, 0.1, 1.1, 0.0)
    assert_same_dets(fresh.RunCascade(_cp(img, 200, 300, 301, 20, 1000, 0.1, 1.1), 0.0), want, "dim 301", Q_TOL_RAW)


def test_queue_overflow_falls_back_to_monolithic(orc, monkeypatch):
    """A survivor queue that is far too small must be detected on the device and answered by the monolithic
    kernel -- same result, no silent truncation."""
    import torch
    from pigo_amd import batch
    monkeypatch.setenv("PIGO_QUEUE_DIV", "100000000")  # queue capacity = the floor ...
    monkeypatch.setenv("PIGO_QUEUE_MIN", "8")           # ... of 8 entries per frame: the per-XCD survivor queues hold 2 entries each
    monkeypatch.setenv("PIGO_SCAN_VARIANT", "2")        # the survivor queue belongs to k_scan_tile / k_tail_deep (variant 3 keeps them for scales > 135 only)
    fresh = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    rows, cols = 540, 960
    f = synth.make_frames("faces", 2, rows, cols, seed=5)
    d_frames = torch.from_numpy(f).to("cuda:0")
    plan = batch.ScanPlan(fresh, rows, cols, max_frames=2, det_cap=256)
    assert plan.info().queue_capacity <= 1024 and plan.info().variant == 2
    dets, counts = plan.alloc_outputs(2)
    plan.run(d_frames, dets, counts)
    torch.cuda.synchronize()
    with pytest.raises(core.PigoError):
        plan.status()  # overflow reported ...
    plan.run(d_frames, dets, counts, sync=True)  # ... and the sync wrapper re-runs with the fallback
    got = batch.dets_to_numpy(dets, counts)
    for k in range(2):
        assert_same_dets(got[k], orc.run_cascade(f[k], rows, cols, cols, 20, 1000, 0.1, 1.1, 0.0), f"fallback frame {k}", Q_TOL_RAW)


def test_batch_detection_overflow_is_reported_not_silent(pg):
    """A frame with more than det_cap detections on the asynchronous batch path: the list is truncated (which records
    survive depends on atomic order), so pigo_plan_status must say so -- PIGO_ERR_CAPACITY -- and d_counts must hold the
    true count."""
    import torch
    from pigo_amd import batch
    f = synth.make_frames("faces", 3, 1080, 1920, seed=1234)
    d_frames = torch.from_numpy(f).cuda()
    plan = batch.ScanPlan(pg, 1080, 1920, max_frames=3, det_cap=16)
    dets, counts = plan.alloc_outputs(3)
    plan.run(d_frames, dets, counts)
    torch.cuda.synchronize()
    with pytest.raises(core.PigoError, match="det_cap"):
        plan.status()
    assert int(counts.max()) > 16
    assert plan.last_flags() == (0, 0, 1)  # not a queue overflow: the caller re-plans with a larger det_cap instead of re-running
    plan.status()  # the flag is cleared by the report
    assert plan.last_flags() == (0, 0, 0)
    big = batch.ScanPlan(pg, 1080, 1920, max_frames=3, det_cap=int(counts.max()))
    d2, c2 = big.alloc_outputs(3)
    big.run(d_frames, d2, c2)
    torch.cuda.synchronize()
    big.status()
    assert torch.equal(c2, counts)


def test_detection_capacity_is_reported(pg):
    f = synth.make_frames("faces", 1, 1080, 1920, seed=1234)[0]
    L = core.load_library()
    import ctypes as C
    out = np.zeros(4, dtype=core.DET_DTYPE)
    n = C.c_int(0)
    pix = np.ascontiguousarray(f).ravel()
    st = L.pigo_run_cascade(pg._need(), pix.ctypes.data, pix.size, 1080, 1920, 1920, 20, 1000, 0.1, 1.1, 0.0, out.ctypes.data, 4, C.byref(n))
    assert st == core.ERR_CAPACITY and n.value > 4


def test_4k_config5_variants_agree_and_match_oracle(pg, orc):
    """BASELINE config 5: 3840x2160, MinSize 20, MaxSize 2000, shift 0.05, scale 1.05 -> 113,382,193 windows."""
    import torch
    from pigo_amd import batch
    f = synth.make_frames("faces", 1, 2160, 3840, seed=1234)
    d_frames = torch.from_numpy(f).to("cuda:0")
    res = {}
    for variant in VARIANTS:
        plan = batch.ScanPlan(pg, 2160, 3840, MinSize=20, MaxSize=2000, ShiftFactor=0.05, ScaleFactor=1.05, max_frames=1, det_cap=32768)
        plan.set_variant(variant)
        assert plan.info().windows_per_frame == 113382193 and plan.info().n_scales == 96
        dets, counts = plan.alloc_outputs(1)
        plan.run(d_frames, dets, counts)
        torch.cuda.synchronize()
        plan.status()  # no queue overflow, i.e. no silent trip through the monolithic fallback, on the step-1 scales
        res[variant] = batch.dets_to_numpy(dets, counts, 0)
    for v in VARIANTS:
        assert_same_dets(res[v], res[2], f"4K v{v} vs v2", Q_TOL_RAW)
    want = orc.run_cascade(f[0], 2160, 3840, 3840, 20, 2000, 0.05, 1.05, 0.0)  # ~15 s of CPU
    assert_same_dets(res[2], want, "4K vs oracle", Q_TOL_RAW)
    wc, ties = orc.cluster_detections(want.copy(), 0.2, want_ties=True)
    got = pg.ClusterDetections(res[2].copy(), 0.2)
    assert_same_dets(got, wc, f"4K clusters (ties={ties})", Q_TOL_RAW)


def test_cpp_mirror_runs_on_gpu(tmp_path):
    """include/pigo.hpp: the C++ mirror of the Go API, as a real program on the GPU (reference test parameters)."""
    import os
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "pigo_amd", "csrc")
    exe = str(tmp_path / "cpp_mirror_check")
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp_mirror_check.cpp"),
                           "-o", exe, "-L" + csrc, "-lpigo_hip", "-Wl,-rpath," + csrc])
    r = subprocess.run([exe, os.path.join(root, "pigo_amd", "data", "facefinder"), os.path.join(root, "pigo_amd", "data", "sample_gray_320x400.bin"),
                        os.path.join(root, "pigo_amd", "data", "puploc")], capture_output=True, text=True)
    assert r.returncode == 0 and "dets=4 clusters=1 (206,154,261," in r.stdout and "gray177=1" in r.stdout and "eye_ok=1" in r.stdout and "wire_ok=1" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_go_order_sort_program(tmp_path):
    """tests/gosort_gpu_check.hip: k_sort_by_q + k_gosort_ties (wave-parallel pdqsort) against the host restatement of Go's
    sort.Slice on 120 random lists -- every tie density, sorted / nearly sorted / descending inputs, lengths on both sides of
    the LDS limit -- with one wave and with the production eight.  core/pigo.go:264-266."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "gosort_gpu_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", os.path.join(root, "tests", "gosort_gpu_check.hip"), "-o", exe])
    for threads, trials in ((512, 12), (64, 3)):
        r = subprocess.run([exe, str(threads), str(trials)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok threads=%d" % threads), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


# ---- the path bench.py times: many 1080p frames, chunked pipeline, side stream, per-XCD queues -----------------------------


def test_rotated_region_kernel_on_frames_with_rotated_faces(pg, orc):
    """The rotated scan of frames whose faces ARE rotated the way angle 0.8 looks for them (the benchmark's upright faces leave a
    rotated scan with a handful of detections per batch): hundreds of survivors per frame go through the rotated region
    kernel's pool, deep lists and the big scales' tail.  12 frames (variant 3: >= 8), every frame against the oracle;
    classifyRotatedRegion, core/pigo.go:150-191."""
    import threading
    import torch
    from scipy import ndimage
    from pigo_amd import batch
    n, rows, cols, angle = 12, 720, 1280, 0.8
    patch = synth.sample_gray()
    rot = {z: ndimage.rotate(ndimage.zoom(patch, z, order=1) if z != 1.0 else patch, -79.0, reshape=True, order=1, mode="nearest") for z in (0.5, 1.0, 1.6)}
    rng = np.random.default_rng(21)
    frames = np.empty((n, rows, cols), dtype=np.uint8)
    for f in range(n):
        bg = synth.syn_noise((rows + 7) // 8, (cols + 7) // 8, seed=77, frame_index=f)
        img = np.repeat(np.repeat(bg, 8, axis=0), 8, axis=1)[:rows, :cols].copy()
        for _ in range(5):
            p = rot[(0.5, 1.0, 1.0, 1.6)[int(rng.integers(0, 4))]]
            ph, pw = p.shape
            if ph > rows or pw > cols:
                continue
            r0, c0 = int(rng.integers(0, rows - ph + 1)), int(rng.integers(0, cols - pw + 1))
            img[r0:r0 + ph, c0:c0 + pw] = p
        frames[f] = img
    want = [None] * n

    def work(f):
        want[f] = orc.run_cascade(frames[f], rows, cols, cols, 20, 700, 0.1, 1.1, angle)

    th = [threading.Thread(target=work, args=(f,)) for f in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert sum(len(w) for w in want) > 100, [len(w) for w in want]  # the rotated faces are found
    plan = batch.ScanPlan(pg, rows, cols, MinSize=20, MaxSize=700, ShiftFactor=0.1, ScaleFactor=1.1, angle=angle, max_frames=n, det_cap=4096)
    assert plan.info().variant == 3
    d_frames = torch.from_numpy(frames).cuda()
    dets, counts = plan.alloc_outputs(n)
    for rep in range(2):
        plan.run(d_frames, dets, counts)
        torch.cuda.synchronize()
        plan.status()
        got = batch.dets_to_numpy(dets, counts)
        for f in range(n):
            assert_same_dets(got[f], want[f], f"rotated faces frame {f} rep {rep}", Q_TOL_RAW)
    plan.set_variant(2)
    plan.run(d_frames, dets, counts)
    torch.cuda.synchronize()
    got = batch.dets_to_numpy(dets, counts)
    for f in range(n):
        assert_same_dets(got[f], want[f], f"rotated faces frame {f} variant 2", Q_TOL_RAW)


@pytest.mark.parametrize("chunks,angle", [(0, 0.0), (4, 0.0), (0, 0.8)])
def test_benchmarked_path_1080p_batch_against_oracle(pg, orc, chunks, angle, monkeypatch):
    """ScanPlan.run + plan.cluster on 40 x 1080p frames (faces + noise mix) exactly as bench.py drives them -- default
    chunking (and 4 chunks), global-gather classes on the side stream, frames dealt to the eight per-XCD survivor queues --
    every frame compared with the oracle (run on host threads), raw lists and clusters, bit-exact.  core/pigo.go:212-308."""
    import threading
    import torch
    from pigo_amd import batch
    monkeypatch.setenv("PIGO_PIPE_CHUNKS", str(chunks))
    n, rows, cols = 40, 1080, 1920
    frames = np.concatenate([synth.make_frames("faces", 34, rows, cols, seed=1234), synth.make_frames("noise", 6, rows, cols, seed=99)])
    d_frames = torch.from_numpy(frames).cuda()
    want, wantc = [None] * n, [None] * n

    def work(lo, hi):
        for f in range(lo, hi):
            want[f] = orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, angle)
            wantc[f] = orc.cluster_detections(want[f].copy(), 0.2, want_ties=True)

    th = [threading.Thread(target=work, args=(k, k + 1)) for k in range(n)]  # ctypes releases the GIL: one core per frame
    for t in th:
        t.start()
    plan = batch.ScanPlan(pg, rows, cols, MinSize=20, MaxSize=1000, ShiftFactor=0.1, ScaleFactor=1.1, angle=angle, max_frames=n, det_cap=1024)
    assert int(plan.info().variant) == 3  # the region kernel: upright AND rotated (landscape frames: clamp-free LDS form)
    dets, counts = plan.alloc_outputs(n)
    cl_out = plan.alloc_cluster_outputs(dets, counts)
    for rep in range(2):  # the second run reuses every queue, counter and event of the first
        plan.run(d_frames, dets, counts)
        sorted_, clusters, ccounts, ties = plan.cluster(dets, counts, 0.2, out=cl_out)
    torch.cuda.synchronize()
    plan.status()
    assert int(counts.max()) <= 1024
    for t in th:
        t.join()
    got = batch.dets_to_numpy(dets, counts)
    cl = batch.dets_to_numpy(clusters, ccounts)
    for f in range(n):
        assert_same_dets(got[f], want[f], f"1080p batch chunks={chunks} angle={angle} frame {f}", Q_TOL_RAW)
        assert int(ties[f]) == wantc[f][1]
        assert_same_dets(cl[f], wantc[f][0], f"1080p batch clusters frame {f} (ties={wantc[f][1]})", Q_TOL_RAW)
    assert sum(len(w) for w in want) > (1000 if angle == 0.0 else 10)


def test_region_deep_list_spill_goes_through_the_tail(pg, orc, monkeypatch):
    """k_scan_region collects the windows alive after tree 27 in a per-region LDS list; a full list spills into
    k_tail_deep's survivor queue.  With the lists cut to 64 entries every face region of these frames spills: the result must
    not change (raw lists bit-exact against the oracle).  core/pigo.go:113-147, :212-258."""
    import threading
    import torch
    from pigo_amd import batch
    monkeypatch.setenv("PIGO_REG_DEEP0", "64")
    monkeypatch.setenv("PIGO_REG_DEEP1", "64")
    n, rows, cols = 8, 1080, 1920
    frames = synth.make_frames("faces", n, rows, cols, seed=4321)
    d_frames = torch.from_numpy(frames).cuda()
    want = [None] * n

    def work(f):
        want[f] = orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, 0.0)

    th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    for t in th:
        t.start()
    plan = batch.ScanPlan(pg, rows, cols, MinSize=20, MaxSize=1000, ShiftFactor=0.1, ScaleFactor=1.1, angle=0.0, max_frames=n, det_cap=1024)
    assert plan.info().variant == 3
    dets, counts = plan.alloc_outputs(n)
    for rep in range(2):
        plan.run(d_frames, dets, counts)
    torch.cuda.synchronize()
    plan.status()
    for t in th:
        t.join()
    got = batch.dets_to_numpy(dets, counts)
    for f in range(n):
        assert_same_dets(got[f], want[f], f"deep-list spill frame {f}", Q_TOL_RAW)
    assert sum(len(w) for w in want) > 200


@pytest.mark.parametrize("env", [{}, {"PIGO_BIG": "0"}, {"PIGO_BIG_CT": "0"}, {"PIGO_BIG_CT": "0", "PIGO_BIG_DEEP_SPLIT": "96"},
                                 {"PIGO_REG_RESERVE0_KB": "0", "PIGO_BIG_FIRST": "1"}, {"PIGO_BIG_POOL_TREE": "2", "PIGO_NH_GLB": "28"},
                                 {"PIGO_BIG_MERGE": "0", "PIGO_REG_COMPRESS": "0"}, {"PIGO_REG_WQ": "64", "PIGO_REG_RESERVE1_KB": "16"}])
def test_big_scales_side_chain_and_its_switches(pg, orc, env, monkeypatch):
    """The scales above the region groups of a variant-3 plan: k_scan_big (persistent, next to the region workgroups) + k_tail_deep
    without an LDS code table.  Forced here: the default; round 3's tile class instead (PIGO_BIG=0); the tail with LDS code
    windows, two launches and five; no LDS reserve with the side chain launched first; pool from tree 2 with the hand-over at
    tree 28; unmerged chunk stages next to an uncompressed region queue; 64-entry region queues with a reserve in both groups.
    Every frame against the oracle, raw lists bit-exact.  core/pigo.go:113-147, :212-258."""
    import threading
    import torch
    from pigo_amd import batch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, rows, cols = 10, 1080, 1920
    frames = np.concatenate([synth.make_frames("faces", 8, rows, cols, seed=4321), synth.make_frames("noise", 2, rows, cols, seed=7)])
    want = [None] * n

    def work(f):
        want[f] = orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, 0.0)

    th = [threading.Thread(target=work, args=(f,)) for f in range(n)]
    for t in th:
        t.start()
    plan = batch.ScanPlan(pg, rows, cols, max_frames=n, det_cap=1024)
    assert int(plan.info().variant) == 3
    d_frames = torch.from_numpy(frames).cuda()
    dets, counts = plan.alloc_outputs(n)
    for rep in range(2):  # (the second run starts from the counters the first one left)
        plan.run(d_frames, dets, counts)
    torch.cuda.synchronize()
    plan.status()
    plan.set_profiling(True)  # the per-kernel timing pass runs the same launches on one stream
    plan.run(d_frames, dets, counts)
    torch.cuda.synchronize()
    names = [k for k, _ in plan.last_timings()]
    plan.set_profiling(False)
    assert ("scan_big" in names) == (env.get("PIGO_BIG") != "0"), names
    if env.get("PIGO_BIG_DEEP_SPLIT") == "96":
        assert "tail_deep5" in names, names
    for t in th:
        t.join()
    got = batch.dets_to_numpy(dets, counts)
    for f in range(n):
        assert_same_dets(got[f], want[f], f"big-scale path {env} frame {f}", Q_TOL_RAW)
    assert sum(len(w) for w in want) > 300


@pytest.mark.parametrize("angle", [0.0, 0.8])
@pytest.mark.parametrize("env", [{"PIGO_REG_QUAD0": "0", "PIGO_REG_QUAD1": "0"}, {"PIGO_REG_QUAD0": "0", "PIGO_REG_QUAD1": "16"},
                                 {"PIGO_REG_QUAD0": "32", "PIGO_REG_QUAD1": "16"},
                                 {"PIGO_REG_QUAD0": "16", "PIGO_REG_QUAD1": "32", "PIGO_NH_REG1": "28", "PIGO_REG_DEEP0": "128", "PIGO_REG_DEEP1": "64"}])
def test_region_deep_list_quad_pass(pg, orc, env, angle, monkeypatch):
    """The deep list of k_scan_region with and without its quad pass (four windows x 16 trees or two x 32 per wave, the float32
    sums running down the segments of lanes, survivors on a second list for the one-window passes): off; 16-tree segments in the
    mid group only; 32 in the small and 16 in the mid group; 16 in the small group and 32 (or what fits) in the mid group with its
    hand-over at tree 28 and lists short enough to spill into k_tail_deep -- each whatever the library's defaults are.  11 frames
    (8 + 3), upright and rotated (faces rotated the way that scan finds them: long deep lists), every frame against the oracle,
    raw lists bit-exact.  core/pigo.go:113-191, :212-258."""
    import threading
    import torch
    from pigo_amd import batch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, rows, cols = 11, 1080, 1920
    frames = np.concatenate([synth.make_frames("faces", 9, rows, cols, seed=97, rotate_deg=(-79.0 if angle else 0.0)),
                             synth.make_frames("noise", 2, rows, cols, seed=11)])
    want = [None] * n

    def work(f):
        want[f] = orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, angle)

    th = [threading.Thread(target=work, args=(f,)) for f in range(n)]
    for t in th:
        t.start()
    plan = batch.ScanPlan(pg, rows, cols, MinSize=20, MaxSize=1000, ShiftFactor=0.1, ScaleFactor=1.1, angle=angle, max_frames=n, det_cap=1024)
    assert int(plan.info().variant) == 3
    d_frames = torch.from_numpy(frames).cuda()
    dets, counts = plan.alloc_outputs(n)
    for rep in range(2):
        plan.run(d_frames, dets, counts)
    torch.cuda.synchronize()
    plan.status()
    for t in th:
        t.join()
    got = batch.dets_to_numpy(dets, counts)
    for f in range(n):
        assert_same_dets(got[f], want[f], f"deep-list quad pass {env} angle {angle} frame {f}", Q_TOL_RAW)
    assert sum(len(w) for w in want) > (300 if angle == 0.0 else 30)


@pytest.mark.parametrize("rccl", [False, True])
def test_sharded_entry_point_world1_matches_plain_path(pg, orc, rccl):
    """pigo_run_batch_sharded (the C ABI a Go / C++ host shards with) at world size 1: scan + cluster + device-side
    packing must give exactly the wire rows of the plain path's lists, padding rows included; raw-list mode as well.
    rccl=True builds a REAL one-rank RCCL communicator (ncclGetUniqueId -> ncclCommInitRank) so that the rows travel through
    ncclAllGather -- the collective an 8-GPU node runs -- instead of the world-1 device copy."""
    import torch
    from pigo_amd import batch, distributed
    n, per, rows, cols, gcap = 11, 16, 270, 480, 8
    frames = synth.make_frames("faces", n, rows, cols, seed=21)
    d_frames = torch.from_numpy(frames).cuda()
    plan = batch.ScanPlan(pg, rows, cols, max_frames=16, det_cap=256)
    dets, counts = plan.alloc_outputs(n)
    plan.run(d_frames, dets, counts)
    _, clusters, ccounts, _ = plan.cluster(dets, counts, 0.2)
    torch.cuda.synchronize()
    comm = distributed.Comm(0, 1, 0, distributed.Comm.unique_id() if rccl else None)
    assert comm.uses_rccl is rccl
    for iou, lists, lcounts in ((0.2, clusters, ccounts), (-1.0, dets, counts)):
        wire = distributed.run_batch_sharded(plan, comm, d_frames, per, iou, gcap)
        torch.cuda.synchronize()
        plan.status()
        ref = distributed.pack_lists(lists, lcounts, gcap, raw_counts=counts)
        assert tuple(wire.shape) == (per, 2 + 4 * gcap)
        assert torch.equal(wire[:n], ref)
        assert int(wire[n:, 0].abs().sum()) == 0 and int(wire[n:, 2:].abs().sum()) == 0 and bool((wire[n:, 1] == distributed.WIRE_PADDING).all())
        host = distributed.pack_lists_host(batch_lists_to_host(lists, n), lcounts.cpu().numpy(), per, gcap)
        assert (host == wire.cpu().numpy()).all()
    want0 = orc.cluster_detections(orc.run_cascade(frames[0], rows, cols, cols, 20, 1000, 0.1, 1.1, 0.0), 0.2)
    got0, cnt0 = distributed.unpack_list_host(distributed.run_batch_sharded(plan, comm, d_frames, per, 0.2, gcap)[0].cpu().numpy(), gcap)
    assert cnt0 == len(want0)
    assert_same_dets(got0, want0[:gcap], "sharded frame 0", Q_TOL_RAW)


def test_sharded_rank_with_a_failing_scan_still_joins_the_collective(pg):
    """A rank whose scan cannot run (here: more frames than the plan holds is refused up front, so provoke the failure inside
    the call with a misaligned frame pointer) must still enqueue its -- zero-count -- rows into the all-gather before it
    reports the error: its peers have already entered ncclAllGather and would hang otherwise (one-rank RCCL communicator)."""
    import torch
    from pigo_amd import batch, distributed
    n, per, rows, cols, gcap = 4, 4, 270, 480, 8
    d_frames = torch.from_numpy(synth.make_frames("faces", n + 1, rows, cols, seed=5)).cuda()
    plan = batch.ScanPlan(pg, rows, cols, max_frames=4, det_cap=256)
    comm = distributed.Comm(0, 1, 0, distributed.Comm.unique_id())
    out = torch.full((per, 2 + 4 * gcap), 7, dtype=torch.int32, device="cuda")
    bad = d_frames.view(-1)[1:1 + n * rows * cols].view(n, rows, cols)  # frame pointer not a multiple of 4: pigo_plan_run refuses
    with pytest.raises(ValueError):
        distributed.run_batch_sharded(plan, comm, bad, per, 0.2, gcap, out=out)
    torch.cuda.synchronize()
    # the collective ran, with padding rows that tell every peer why: PIGO_WIRE_RANK_FAILED
    assert int(out[:, 0].abs().sum()) == 0 and int(out[:, 2:].abs().sum()) == 0
    assert bool((out[:, 1] == (distributed.WIRE_RANK_FAILED | distributed.WIRE_PADDING)).all())
    wire = distributed.run_batch_sharded(plan, comm, d_frames[:n], per, 0.2, gcap, out=out)  # the communicator is still usable
    torch.cuda.synchronize()
    plan.status()
    assert int(wire[:, 0].sum()) > 0


def test_tuning_switches_need_pigo_tuning(pg, monkeypatch):
    """A stray tuning variable in a production environment must not move a plan onto a path nobody benchmarks: without
    PIGO_TUNING=1 the library ignores PIGO_BIG=0 (k_scan_big still runs); PIGO_SCAN_VARIANT is a user switch and always works."""
    import torch
    from pigo_amd import batch
    monkeypatch.delenv("PIGO_TUNING", raising=False)
    monkeypatch.setenv("PIGO_BIG", "0")
    plan = batch.ScanPlan(pg, 1080, 1920, max_frames=8, det_cap=256)
    assert int(plan.info().variant) == 3
    d_frames = torch.from_numpy(synth.make_frames("faces", 8, 1080, 1920, seed=3)).cuda()
    dets, counts = plan.alloc_outputs(8)
    plan.set_profiling(True)
    plan.run(d_frames, dets, counts)
    torch.cuda.synchronize()
    assert "scan_big" in [k for k, _ in plan.last_timings()]
    monkeypatch.setenv("PIGO_SCAN_VARIANT", "2")
    assert int(batch.ScanPlan(pg, 1080, 1920, max_frames=8, det_cap=256).info().variant) == 2


def test_comm_abort_and_init_deadline(pg, monkeypatch):
    """pigo_comm_abort (ncclCommAbort) on a real one-rank RCCL communicator: the handle refuses further collectives and can
    still be destroyed; and pigo_comm_init's deadline: a rank whose peer never calls ncclCommInitRank gets PIGO_ERR_HIP back
    after PIGO_COMM_INIT_TIMEOUT_S instead of hanging (world 2, only rank 0 shows up)."""
    import time
    import torch
    from pigo_amd import batch, distributed
    n, rows, cols, gcap = 4, 270, 480, 8
    d_frames = torch.from_numpy(synth.make_frames("faces", n, rows, cols, seed=5)).cuda()
    plan = batch.ScanPlan(pg, rows, cols, max_frames=n, det_cap=256)
    comm = distributed.Comm(0, 1, 0, distributed.Comm.unique_id())
    assert comm.uses_rccl
    wire = distributed.run_batch_sharded(plan, comm, d_frames, n, 0.2, gcap)
    torch.cuda.synchronize()
    plan.status()
    assert int(wire[:, 0].sum()) > 0
    comm.abort()
    with pytest.raises(ValueError):
        distributed.run_batch_sharded(plan, comm, d_frames, n, 0.2, gcap)
    del comm
    # The deadline, in a process of its own: the timed-out rank's helper thread stays blocked inside ncclCommInitRank until its
    # process ends, so the child reports and leaves with os._exit -- nothing of it survives into this suite.
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, time\n"
            "sys.path.insert(0, %r)\n"
            "os.environ['PIGO_COMM_INIT_TIMEOUT_S'] = '3'\n"
            "from pigo_amd import core, distributed\n"
            "uid = distributed.Comm.unique_id()\n"
            "t0 = time.time()\n"
            "try:\n"
            "    distributed.Comm(0, 2, 0, uid)\n"
            "    print('NO_ERROR', flush=True)\n"
            "except core.PigoError as e:\n"
            "    print('DEADLINE %%.2f %%s' %% (time.time() - t0, e), flush=True)\n"
            "os._exit(0)\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith(("DEADLINE", "NO_ERROR"))]
    assert r.returncode == 0 and line and line[0].startswith("DEADLINE"), (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    waited = float(line[0].split()[1])
    assert 2.0 < waited < 60.0, line[0]  # PIGO_ERR_HIP after ~3 s instead of hanging: world 2, only rank 0 ever shows up


def batch_lists_to_host(lists, n):
    return lists[:n].cpu().numpy().view(core.DET_DTYPE).reshape(n, lists.shape[1])


@pytest.mark.parametrize("graph", ["0", "1"])
def test_run_cascade_is_reentrant_four_threads_one_handle(orc, graph, monkeypatch):
    """The reference's RunCascade is re-entrant (examples/web/main.go:71,141 share one *Pigo between request handlers): four
    host threads hammer ONE handle with different frames and parameters; every result must be the oracle's.  (Each call
    takes its own slot -- plan, buffers, stream, captured graph -- so the calls overlap on the GPU instead of queueing.)"""
    import threading
    monkeypatch.setenv("PIGO_GRAPH_FRAMES", graph)  # with and without the captured upload-scan-download graph per slot
    pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    jobs = []
    for k in range(4):
        rows, cols = (270, 480) if k % 2 == 0 else (400, 320)
        img = synth.syn_faces(rows, cols, seed=40 + k) if k % 2 == 0 else synth.sample_gray()
        shift = 0.1 if k < 2 else 0.2
        jobs.append((img, rows, cols, shift, orc.run_cascade(img, rows, cols, cols, 20, 1000, shift, 1.1, 0.0)))
    errors = []

    def work(k):
        img, rows, cols, shift, want = jobs[k]
        try:
            for rep in range(12):
                got = pg.RunCascade(_cp(img, rows, cols, cols, 20, 1000, shift, 1.1), 0.0)
                assert_same_dets(got, want, f"thread {k} rep {rep}", Q_TOL_RAW)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    # two threads, SAME parameters: two slots of one key
    errors.clear()
    th = [threading.Thread(target=work, args=(0,)) for _ in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_cluster_detections_is_reentrant_four_threads_one_handle(orc):
    """The reference's ClusterDetections is a pure function (core/pigo.go:262-308) that goroutines call concurrently on one
    *Pigo (examples/web/main.go:141-144).  Until round 5 pigo_cluster_detections held the handle's mutex across its launch and
    synchronisation; now a call owns a slot (pinned staging, stream, device scratch) and the mutex covers the slot list only.
    Four threads, one handle, lists of every path's length (one-launch pinned path, k_cluster on the device, seeds / members /
    compact), each thread's results against the oracle's, next to threads that call RunCascade on the same handle."""
    import threading
    fresh = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    rng = np.random.default_rng(99)

    def boxes(n, spread):
        return [(int(rng.integers(50, spread)), int(rng.integers(50, spread)), int(rng.integers(20, 120)), np.float32(rng.random() * 30 + 0.01 * i)) for i in range(n)]

    lists = [boxes(40, 600), boxes(700, 1500), boxes(1500, 3000), boxes(3000, 4000), boxes(9, 200)]
    ious = [0.0, 0.1, 0.2, 0.15, 0.01]
    wants = [orc.cluster_detections(oracle.make_dets(l), iou) for l, iou in zip(lists, ious)]
    img = synth.syn_faces(240, 320, seed=11)
    want_scan = orc.run_cascade(img, 240, 320, 320, 20, 1000, 0.1, 1.1, 0.0)
    errors = []

    def work(k):
        try:
            for rep in range(12):
                j = (k + rep) % len(lists)
                got = fresh.ClusterDetections(core.make_dets(lists[j]), ious[j])
                assert_same_dets(got, wants[j], f"thread {k} rep {rep} list {j}", Q_TOL_RAW)
                if k == 3:
                    assert_same_dets(fresh.RunCascade(_cp(img, 240, 320, 320, 20, 1000, 0.1, 1.1), 0.0), want_scan, f"RunCascade next to clustering, rep {rep}", Q_TOL_RAW)
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {k}: {e!r}")

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_one_frame_plans_in_a_process_full_of_streams(pg, orc):
    """A process's HIP streams share a handful of hardware queues.  Until round 4 a plan of a few frames forked onto a side
    stream of its own and had to probe it against the caller's; now such a plan is ONE launch on the caller's stream
    (k_scan_one).  Twelve live streams and a live batch plan: one-frame plans run on the default stream and on two of those
    streams, several times each, and a fresh handle serves RunCascade -- every result the oracle's.  core/pigo.go:212-258."""
    import torch
    from pigo_amd import batch
    rows, cols = 480, 640
    streams = [torch.cuda.Stream() for _ in range(12)]
    frames = synth.make_frames("faces", 8, rows, cols, seed=77)
    d_frames = torch.from_numpy(frames).cuda()
    want = [orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, 0.0) for f in range(8)]
    big = batch.ScanPlan(pg, rows, cols, max_frames=8, det_cap=1024)
    bd, bc = big.alloc_outputs(8)
    big.run(d_frames, bd, bc)
    torch.cuda.synchronize()
    big.status()
    got8 = batch.dets_to_numpy(bd, bc)
    for f in range(8):
        assert_same_dets(got8[f], want[f], f"batch plan frame {f}", Q_TOL_RAW)
    plans = [batch.ScanPlan(pg, rows, cols, max_frames=1, det_cap=1024) for _ in range(3)]
    for k, plan in enumerate(plans):
        dets, counts = plan.alloc_outputs(1)
        for st in (None, streams[3 * k + 1], streams[3 * k + 2], None):
            for rep in range(2):
                f = (k + rep) % 8
                if st is None:
                    plan.run(d_frames[f:f + 1], dets, counts)
                else:
                    st.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(st):
                        plan.run(d_frames[f:f + 1], dets, counts, stream=st)
                torch.cuda.synchronize()
                plan.status()
                assert_same_dets(batch.dets_to_numpy(dets, counts)[0], want[f], f"one-frame plan {k} stream {st} rep {rep}", Q_TOL_RAW)
    pg2 = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    for f in range(3):
        got = pg2.RunCascade(_cp(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1), 0.0)
        assert_same_dets(got, want[f], f"RunCascade next to {len(streams)} streams, frame {f}", Q_TOL_RAW)
    assert sum(len(w) for w in want) > 20


@pytest.mark.parametrize("nframes,angle,kind", [(1, 0.0, "faces"), (3, 0.0, "faces"), (1, 0.8, "faces"), (7, 0.0, "noise")])
def test_one_launch_plans_hand_off_under_uneven_load(pg, orc, nframes, angle, kind):
    """Plans of fewer than 8 frames run as ONE persistent launch (k_scan_one: regions and big-scale chunks as items, survivors
    handed to consumer waves of the same grid through global queues, the last workgroup restores the reference's order).  Its
    in-launch hand-offs are exercised the way they fail: hundreds of launches NEXT TO a batch plan that keeps the chip busy on
    another stream (workgroups of the launch start late and unevenly), every result compared with the first -- itself the
    oracle's, record for record.  Two bugs of exactly this kind were found this way in round 5 (a consumer's claim overtaking
    its read of the finished-items counter; a poison entry left for the next launch).  core/pigo.go:212-258."""
    import torch
    from pigo_amd import batch
    rows, cols = 720, 1280
    frames = synth.make_frames(kind, nframes, rows, cols, seed=31)
    plan = batch.ScanPlan(pg, rows, cols, angle=angle, max_frames=nframes, det_cap=2048)
    assert plan.info().variant == 3
    dev = torch.from_numpy(frames).cuda()
    dets, counts = plan.alloc_outputs(nframes)
    plan.run(dev, dets, counts)
    torch.cuda.synchronize()
    plan.status()
    got = batch.dets_to_numpy(dets, counts)
    for f in range(nframes):
        want = orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, angle)
        assert_same_dets(got[f], want, f"k_scan_one frame {f}", Q_TOL_RAW)
    ref_d, ref_c = dets.clone(), counts.clone()
    big = batch.ScanPlan(pg, rows, cols, max_frames=32, det_cap=1024)
    bf = torch.from_numpy(synth.make_frames("faces", 32, rows, cols, seed=5)).cuda()
    bd, bc = big.alloc_outputs(32)
    side = torch.cuda.Stream()
    for i in range(400):
        if i % 8 == 0:
            big.run(bf, bd, bc, stream=side)
        dets.zero_()
        plan.run(dev, dets, counts)
        torch.cuda.current_stream().synchronize()
        plan.status()
        assert torch.equal(counts, ref_c) and torch.equal(dets, ref_d), f"launch {i} next to a busy batch plan differs from the first"
    torch.cuda.synchronize()
    big.status()


def test_one_launch_plans_next_to_another_process(pg, orc):
    """The hand-offs of k_scan_one with ANOTHER PROCESS keeping the device busy (its 64-frame batch plan in a loop): the launch's
    workgroups get the chip in pieces and late.  Every one of 300 launches must return the first one's lists -- the oracle's --
    and none may give up a hand-off (PIGO_ERR_TIMEOUT; the wait is bounded by wall-clock time, PIGO_ONE_TIMEOUT_MS, since
    round 6 -- 2^17 polls before, which a descheduled launch can exceed while perfectly healthy).  core/pigo.go:212-258."""
    import subprocess
    import sys
    import time
    import torch
    from pigo_amd import batch
    rows, cols = 720, 1280
    hog = subprocess.Popen([sys.executable, "-c", (
        "import os, sys, time\n"
        "sys.path.insert(0, os.getcwd())\n"
        "import torch\n"
        "from pigo_amd import batch, core, synth\n"
        "pg = core.NewPigo(0).Unpack(synth.facefinder_bytes())\n"
        "plan = batch.ScanPlan(pg, 720, 1280, max_frames=64, det_cap=1024)\n"
        "fr = torch.from_numpy(synth.make_frames('faces', 64, 720, 1280, seed=5)).cuda()\n"
        "d, c = plan.alloc_outputs(64)\n"
        "print('hog ready', flush=True)\n"
        "t0 = time.time()\n"
        "n = 0\n"
        "while time.time() - t0 < float(sys.argv[1]):\n"
        "    for _ in range(8):\n"
        "        plan.run(fr, d, c)\n"
        "    torch.cuda.synchronize()\n"
        "    n += 8\n"
        "plan.status()\n"
        "print('hog steps', n, flush=True)\n"), "12"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
        cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        seen = ""
        for _ in range(50):  # (the runtime may print a line or two of its own first)
            line = hog.stdout.readline()
            seen += line
            if "hog ready" in line or not line:
                break
        assert "hog ready" in seen, seen + hog.stdout.read()
        for nframes, angle in ((1, 0.0), (3, 0.0)):
            frames = synth.make_frames("faces", nframes, rows, cols, seed=31)
            plan = batch.ScanPlan(pg, rows, cols, angle=angle, max_frames=nframes, det_cap=2048)
            assert plan.info().variant == 3
            dev = torch.from_numpy(frames).cuda()
            dets, counts = plan.alloc_outputs(nframes)
            plan.run(dev, dets, counts)
            torch.cuda.synchronize()
            plan.status()
            got = batch.dets_to_numpy(dets, counts)
            for f in range(nframes):
                assert_same_dets(got[f], orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, angle), f"frame {f} next to the other process", Q_TOL_RAW)
            ref_d, ref_c = dets.clone(), counts.clone()
            for i in range(150):
                assert hog.poll() is None, "the other process ended early: " + hog.stdout.read()
                dets.zero_()
                plan.run(dev, dets, counts)
                torch.cuda.current_stream().synchronize()
                plan.status()  # raises on PIGO_ERR_TIMEOUT / a queue flag
                assert torch.equal(counts, ref_c) and torch.equal(dets, ref_d), f"launch {i} ({nframes} frames) next to another process differs from the first"
    finally:
        try:
            out, _ = hog.communicate(timeout=60)
        except subprocess.TimeoutExpired:
            hog.kill()
            out = ""
    assert "hog steps" in out, out


def _small_face(f, crop=None):
    """The sample face shrunk by the integer factor f (box average), optionally cropped: faces of scale ~24...40 in frames of a
    few regions."""
    g = synth.sample_gray()
    H, W = (400 // f) * f, (320 // f) * f
    im = g[:H, :W].reshape(H // f, f, W // f, f).mean(axis=(1, 3)).round().astype(np.uint8)
    if crop is not None:
        im = im[crop[0]:crop[1], crop[2]:crop[3]]
    return np.ascontiguousarray(im)


@pytest.mark.parametrize("angle", [0.0, 0.03])
def test_one_launch_plan_on_frames_of_fewer_than_eight_items(orc, angle):
    """A frame of a few dozen pixels is fewer than eight items for k_scan_one, whose producers spread survivors over all eight
    global queues while a workgroup only drains queue blockIdx & 7: with a grid of `nitems` workgroups the faces whose windows
    landed in the queues beyond it were dropped without a word, and their entries stayed behind for the plan's next run (advisor,
    round 5).  The launch has at least eight workgroups now.  Every frame is scanned twice through one handle (the second run
    starts from whatever the first left in the queues), min size 16 so that the small faces are inside the ladder;
    core/pigo.go:212-258."""
    fresh = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    total = 0
    for f, crop, shift in ((10, None, 0.05), (10, (2, 34, 0, 32), 0.1), (9, (4, 44, 0, 32), 0.05), (8, None, 0.1), (9, None, 0.05), (6, None, 0.05)):
        im = _small_face(f, crop)
        rows, cols = im.shape
        dim = (cols + 3) & ~3
        buf = np.full((rows, dim), 128, dtype=np.uint8)
        buf[:, :cols] = im
        want = orc.run_cascade(buf, rows, cols, dim, 16, 1000, shift, 1.1, angle)
        for rep in range(2):
            got = fresh.RunCascade(_cp(buf, rows, cols, dim, 16, 1000, shift, 1.1), angle)
            assert_same_dets(got, want, f"{rows}x{cols} (dim {dim}) shift {shift} angle {angle}, run {rep}", Q_TOL_RAW)
        total += len(want)
    assert total > 100 or angle > 0.0


def test_one_launch_plan_with_a_list_too_long_for_its_own_order_restore(pg, orc):
    """k_scan_one's last workgroup ranks the detections out of its LDS (<= 16,384 per frame); a plan with a larger det_cap keeps
    k_restore_order behind the launch and zeroes the launch's counters from the host instead -- the other half of launch_scan's
    one-launch branch.  Two frames, twice (the second run starts from what the first left)."""
    import torch
    from pigo_amd import batch
    rows, cols = 480, 640
    frames = synth.make_frames("faces", 2, rows, cols, seed=21)
    plan = batch.ScanPlan(pg, rows, cols, max_frames=2, det_cap=20000)
    assert plan.info().variant == 3
    dev = torch.from_numpy(frames).cuda()
    dets, counts = plan.alloc_outputs(2)
    for _ in range(2):
        dets.zero_()
        plan.run(dev, dets, counts)
        torch.cuda.synchronize()
        plan.status()
        got = batch.dets_to_numpy(dets, counts)
        for f in range(2):
            assert_same_dets(got[f], orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, 0.0), f"det_cap 20000, frame {f}", Q_TOL_RAW)


def test_one_launch_plan_reports_a_queue_overflow(pg, monkeypatch):
    """k_scan_one's global queues are sized for 1/32 of a frame's windows each; a frame that keeps more alive than that must
    raise the plan's queue flag (PIGO_ERR_CAPACITY from pigo_plan_status), not hang and not drop windows silently -- and the
    plan must be usable again afterwards (the status call clears queues and counters).  Forced here with a tiny queue."""
    import torch
    from pigo_amd import batch
    monkeypatch.setenv("PIGO_TUNING", "1")
    monkeypatch.setenv("PIGO_ONE_QCAP", "8")
    rows, cols = 480, 640
    frames = synth.make_frames("faces", 1, rows, cols, seed=9)
    plan = batch.ScanPlan(pg, rows, cols, max_frames=1, det_cap=1024)
    monkeypatch.delenv("PIGO_ONE_QCAP")
    assert plan.info().variant == 3
    dev = torch.from_numpy(frames).cuda()
    dets, counts = plan.alloc_outputs(1)
    plan.run(dev, dets, counts)
    torch.cuda.synchronize()
    with pytest.raises(core.PigoError):
        plan.status()
    assert plan.last_flags()[0] != 0
    ok = batch.ScanPlan(pg, rows, cols, max_frames=1, det_cap=1024)
    d2, c2 = ok.alloc_outputs(1)
    ok.run(dev, d2, c2)
    torch.cuda.synchronize()
    ok.status()
    assert int(c2[0]) > 0
    # RunCascade on such a frame falls back to the kernel without a queue and still returns the reference's list
    monkeypatch.setenv("PIGO_ONE_QCAP", "8")
    pg2 = core.NewPigo(0).Unpack(synth.facefinder_bytes())
    got = pg2.RunCascade(_cp(frames[0], rows, cols, cols, 20, 1000, 0.1, 1.1), 0.0)
    assert_same_dets(got, batch.dets_to_numpy(d2, c2)[0], "RunCascade behind an overflowing k_scan_one queue", Q_TOL_RAW)


@pytest.mark.parametrize("runs", [False] + ([True] if __import__("os").environ.get("PIGO_STRESS_FULL") else []))
def test_slot_capture_next_to_plan_builds_on_other_handles(orc, runs, monkeypatch):
    """Round 2's abort, root cause pinned with rocgdb (gpurun_out/r3/abort_gdb_*.txt): while one thread captures the
    upload-scan-download graph of a new RunCascade slot (hipStreamBeginCapture, thread-local mode), another thread's
    plan_build called hipDeviceSynchronize() -- refused "when stream is capturing" even from a foreign thread, and the capture
    is invalidated on the way.  plan_build now stays on a private stream (no device-wide call, nothing on the null stream), so
    slot builds, plan builds and hipMalloc-heavy calls on OTHER handles may run next to a capture without a process lock.
    (runs=True -- only with PIGO_STRESS_FULL=1 -- also scans batches on those other plans from the legacy null stream while
    the capture is going on; the round-3 stress script that drove the variations of that is gone with its round.)"""
    import threading
    import torch
    from pigo_amd import batch
    monkeypatch.setenv("PIGO_GRAPH_FRAMES", "1")
    packet = synth.facefinder_bytes()
    imgs = [synth.syn_faces(120 + 8 * k, 160 + 4 * k, seed=70 + k) for k in range(6)]
    wants = [orc.run_cascade(im, im.shape[0], im.shape[1], im.shape[1], 20, 1000, 0.1, 1.1, 0.0) for im in imgs]
    errors, stop = [], threading.Event()

    def capture_slots():  # every call has new parameters: a new slot, a new plan, a new captured graph
        try:
            pg = core.NewPigo(0).Unpack(packet)
            for rep in range(3):
                for k, im in enumerate(imgs):
                    got = pg.RunCascade(_cp(im, im.shape[0], im.shape[1], im.shape[1], 20 + rep, 1000, 0.1, 1.1), 0.0)
                    if rep == 0:
                        assert_same_dets(got, wants[k], f"capture thread frame {k}", Q_TOL_RAW)
        except Exception as e:  # noqa: BLE001
            errors.append(("capture", repr(e)))
        finally:
            stop.set()

    def build_plans(seed):  # a different handle: plan builds (tables, uploads, ~20 allocations each) and batch runs
        try:
            pg = core.NewPigo(0).Unpack(packet)
            k = 0
            while not stop.is_set() or k < 4:
                rows, cols = 96 + 8 * ((k + seed) % 5), 128 + 4 * ((k + seed) % 7)
                plan = batch.ScanPlan(pg, rows, cols, max_frames=8, det_cap=256)  # ~25 hipMalloc, 3 table-build launches, uploads
                if runs:
                    fr = torch.from_numpy(synth.make_frames("faces", 8, rows, cols, seed=seed + k)).cuda()
                    dets, counts = plan.alloc_outputs(8)
                    plan.run(fr, dets, counts, sync=True)
                del plan                                                          # ... and as many hipFree (each a device-wide wait)
                k += 1
                if k > 60:
                    break
        except Exception as e:  # noqa: BLE001
            errors.append(("build", repr(e)))

    th = [threading.Thread(target=capture_slots)] + [threading.Thread(target=build_plans, args=(s,)) for s in (1, 2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_small_batch_graph_replay_matches_and_survives_buffer_changes(pg, orc, monkeypatch):
    """pigo_plan_run replays a captured graph for small batches; new buffers or a new batch size must re-capture, and the
    results must not depend on whether the launch sequence was replayed or issued."""
    import torch
    from pigo_amd import batch
    rows, cols = 270, 480
    frames = synth.make_frames("faces", 3, rows, cols, seed=9)
    want = [orc.run_cascade(frames[f], rows, cols, cols, 20, 1000, 0.1, 1.1, 0.0) for f in range(3)]
    d_frames = torch.from_numpy(frames).cuda()
    monkeypatch.setenv("PIGO_GRAPH_FRAMES", "4")  # (read at plan creation; off by default)
    plan = batch.ScanPlan(pg, rows, cols, max_frames=3, det_cap=512)
    for n in (1, 3, 2, 3):
        dets, counts = plan.alloc_outputs(n)  # fresh buffers every round: the cached graph must not be reused blindly
        for rep in range(3):
            plan.run(d_frames[:n], dets, counts)
        torch.cuda.synchronize()
        plan.status()
        got = batch.dets_to_numpy(dets, counts)
        for f in range(n):
            assert_same_dets(got[f], want[f], f"graph replay n={n} frame {f}", Q_TOL_RAW)
