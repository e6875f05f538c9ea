"""The face -> eyes -> 15 landmark points sequence of core/flploc_test.go:84-153, written once against an abstract
backend so that the oracle, the NumPy restatement and the HIP path run exactly the same steps.

A backend offers
    run_detector(cascade_name, row, col, scale, perturbs, flip_v, rnd, pool) -> (row, col, scale)
    get_landmark_point(cascade_name, left_eye, right_eye, perturb, flip_v, rnd, pool) -> (row, col, scale)
with eyes as (row, col, scale) triples.  One pool object is threaded through the whole sequence: a single-goroutine Go
program gets the same sync.Pool object back on every Get (no GC in between), so the stale entries that the reference
sorts along with the fresh ones (core/puploc.go:267-269) are those of the previous call.
"""
import numpy as np

from pigo_amd import synth

EYE_CASCADES = ["lp46", "lp44", "lp42", "lp38", "lp312"]   # core/flploc_test.go:86
MOUTH_CASCADES = ["lp93", "lp84", "lp82", "lp81"]          # core/flploc_test.go:87
PERTURB = 63                                                # core/flploc_test.go:13


def eye_requests(det_row, det_col, det_scale):
    """core/flploc_test.go:110-125: the two RunDetector starting points of a face (float32 arithmetic like the Go text)."""
    f = np.float32
    left = (det_row - int(f(0.075) * f(det_scale)), det_col - int(f(0.175) * f(det_scale)), float(f(det_scale) * f(0.25)), 50)
    right = (det_row - int(f(0.075) * f(det_scale)), det_col + int(f(0.185) * f(det_scale)), float(f(det_scale) * f(0.25)), 50)
    return left, right


def run_sequence(backend, face, seed=1234, pool=None):
    """-> dict with the two eyes and the 15 landmark points [(name, flip, (row, col, scale)), ...]"""
    pool = np.zeros((3, 63), np.float32) if pool is None else pool
    draw = iter(range(1 << 30))

    def rnd():
        return synth.syn_uniform32(3 * 63, seed=seed, index=next(draw))

    (lr, lc, ls, lp), (rr, rc, rs, rp) = eye_requests(*face)
    left = backend.run_detector("puploc", lr, lc, ls, lp, False, rnd(), pool)
    right = backend.run_detector("puploc", rr, rc, rs, rp, False, rnd(), pool)
    pts = []
    for name in EYE_CASCADES:           # core/flploc_test.go:127-138
        for flip in (False, True):
            pts.append((name, flip, backend.get_landmark_point("lps/" + name, left, right, PERTURB, flip, rnd(), pool)))
    for name in MOUTH_CASCADES:         # core/flploc_test.go:139-146
        pts.append((name, False, backend.get_landmark_point("lps/" + name, left, right, PERTURB, False, rnd(), pool)))
    pts.append(("lp84", True, backend.get_landmark_point("lps/lp84", left, right, PERTURB, True, rnd(), pool)))  # :148
    return {"left": left, "right": right, "points": pts}


class OracleBackend:
    def __init__(self, gray, rows, cols, dim):
        import oracle
        self.o, self.casc = oracle, {}
        self.img = (gray, rows, cols, dim)

    def _c(self, name):
        if name not in self.casc:
            self.casc[name] = self.o.OraclePuploc.unpack(synth.cascade_bytes(name))
        return self.casc[name]

    def run_detector(self, name, row, col, scale, perturbs, flip_v, rnd, pool):
        g, rows, cols, dim = self.img
        return self._c(name).run_detector(row, col, scale, perturbs, g, rows, cols, dim, 0.0, flip_v, rnd, pool)

    def get_landmark_point(self, name, left, right, perturb, flip_v, rnd, pool):
        g, rows, cols, dim = self.img
        return self._c(name).get_landmark_point(left, right, g, rows, cols, dim, perturb, flip_v, rnd, pool)


class NumpyBackend:
    """Independent check: GetLandmarkPoint's float64 prologue (core/flploc.go:37-51) is restated here as well."""

    def __init__(self, gray, rows, cols, dim):
        from oracle.np_restatement import NpPuploc
        self.cls, self.casc = NpPuploc, {}
        self.img = (gray, rows, cols, dim)

    def _c(self, name):
        if name not in self.casc:
            self.casc[name] = self.cls(synth.cascade_bytes(name))
        return self.casc[name]

    def run_detector(self, name, row, col, scale, perturbs, flip_v, rnd, pool):
        g, rows, cols, dim = self.img
        return self._c(name).run_detector(row, col, scale, perturbs, g, rows, cols, dim, 0.0, flip_v, rnd, pool)

    def get_landmark_point(self, name, left, right, perturb, flip_v, rnd, pool):
        dx = (left[0] - right[0]) ** 2
        dy = (left[1] - right[1]) ** 2
        dist = float(np.sqrt(np.float64(dx + dy)))
        row = (left[0] + right[0]) / 2.0 + 0.25 * dist
        col = (left[1] + right[1]) / 2.0 + 0.15 * dist
        return self.run_detector(name, int(row), int(col), float(np.float32(3.0 * dist)), perturb, flip_v, rnd, pool)


class HipBackend:
    """The product through the reference-shaped Python mirror (pigo_amd.core.PuplocCascade)."""

    def __init__(self, gray, rows, cols, dim):
        from pigo_amd import core
        self.core, self.casc = core, {}
        self.img = core.ImageParams(Pixels=gray, Rows=rows, Cols=cols, Dim=dim)

    def _c(self, name):
        if name not in self.casc:
            self.casc[name] = self.core.NewPuplocCascade(0).UnpackCascade(synth.cascade_bytes(name))
        return self.casc[name]

    def run_detector(self, name, row, col, scale, perturbs, flip_v, rnd, pool):
        p = self._c(name).RunDetector(self.core.Puploc(row, col, scale, perturbs), self.img, 0.0, flip_v, rnd=rnd, pool=pool)
        return p.Row, p.Col, np.float32(p.Scale)

    def get_landmark_point(self, name, left, right, perturb, flip_v, rnd, pool):
        P = self.core.Puploc
        p = self._c(name).GetLandmarkPoint(P(left[0], left[1], float(left[2])), P(right[0], right[1], float(right[2])), self.img, perturb, flip_v,
                                           rnd=rnd, pool=pool)
        return p.Row, p.Col, np.float32(p.Scale)
