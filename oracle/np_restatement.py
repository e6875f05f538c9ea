"""Independent NumPy restatement of the reference hot path -- the oracle's cross-check.

TEST INFRASTRUCTURE (same rules as oracle/pigo_oracle.c: tests / smoke / cpu_baseline only).

Written separately from the C oracle and in a different shape (vectorised over all windows of one
scale, tree by tree) so that a transcription slip in either restatement shows up as a diff:

    Unpack                 /root/reference/core/pigo.go:51-110
    classifyRegion         /root/reference/core/pigo.go:113-147
    classifyRotatedRegion  /root/reference/core/pigo.go:150-191
    RunCascade             /root/reference/core/pigo.go:212-258
    ClusterDetections      /root/reference/core/pigo.go:262-308   (stable tie order -- see below)

float32 leaf accumulation is done with numpy float32 arrays (one IEEE add per tree, sequential in
tree order), the ladder in Python floats (IEEE double), the IoU in Python floats.

Tie rule: ``cluster_detections`` here sorts STABLY by Q.  Go's sort.Slice is unstable for n > 12, so
this function agrees with the C oracle (which restates Go's pdqsort) only on tie-free inputs or
n <= 12; tests use it exactly that way.
"""
import struct

import numpy as np

Q_COS = [256, 251, 236, 212, 181, 142, 97, 49, 0, -49, -97, -142, -181, -212, -236, -251, -256, -251, -236, -212, -181, -142,
         -97, -49, 0, 49, 97, 142, 181, 212, 236, 251, 256]  # pigo.go:156
Q_SIN = [0, 49, 97, 142, 181, 212, 236, 251, 256, 251, 236, 212, 181, 142, 97, 49, 0, -49, -97, -142, -181, -212, -236, -251,
         -256, -251, -236, -212, -181, -142, -97, -49, 0]  # pigo.go:157


class NpPigo:
    def __init__(self, codes, preds, thr, depth):
        self.codes = codes  # int8 [ntrees, 4*2^depth] (4 leading zero bytes per tree, pigo.go:79)
        self.preds = preds  # float32 [ntrees, 2^depth]
        self.thr = thr  # float32 [ntrees]
        self.depth = depth
        self.ntrees = len(thr)

    @classmethod
    def unpack(cls, packet: bytes):
        depth, ntrees = struct.unpack_from("<II", packet, 8)  # pigo.go:61-68
        nleaf = 1 << depth
        ncode = 4 * nleaf - 4
        rec = ncode + 4 * nleaf + 4
        if len(packet) < 16 + ntrees * rec:
            raise IndexError("packet too short")
        body = np.frombuffer(packet, dtype=np.uint8, count=ntrees * rec, offset=16).reshape(ntrees, rec)
        codes = np.zeros((ntrees, 4 * nleaf), dtype=np.int8)
        codes[:, 4:] = body[:, :ncode].view(np.int8)
        preds = np.ascontiguousarray(body[:, ncode:ncode + 4 * nleaf]).view("<f4").reshape(ntrees, nleaf)
        thr = np.ascontiguousarray(body[:, ncode + 4 * nleaf:]).view("<f4").reshape(ntrees)
        return cls(codes, preds.copy(), thr.copy(), depth)

    # one scale, all windows at once ---------------------------------------------------------------
    def _scan_scale(self, pix, npix, rows, cols, dim, s, step, angle):
        off = s // 2 + 1 if s >= 0 else -((-s) // 2) + 1  # Go int division truncates toward zero
        rr = np.arange(off, rows - off + 1, step, dtype=np.int64)
        cc = np.arange(off, cols - off + 1, step, dtype=np.int64)
        if len(rr) == 0 or len(cc) == 0:
            return np.zeros((0, 4)), 0
        R, Cc = np.meshgrid(rr, cc, indexing="ij")
        R = R.ravel()
        Cc = Cc.ravel()
        nwin = R.size
        alive = np.arange(nwin)
        out = np.zeros(nwin, dtype=np.float32)
        nleaf = 1 << self.depth
        rotated = angle > 0.0
        if rotated:
            k = int(32.0 * min(angle, 1.0))
            qsin, qcos = s * Q_SIN[k], s * Q_COS[k]
        for t in range(self.ntrees):
            if alive.size == 0:
                break
            r, c = R[alive], Cc[alive]
            idx = np.ones(alive.size, dtype=np.int64)
            tc = self.codes[t].astype(np.int64)
            for _ in range(self.depth):
                c0, c1, c2, c3 = tc[4 * idx], tc[4 * idx + 1], tc[4 * idx + 2], tc[4 * idx + 3]
                if not rotated:
                    x1 = ((r * 256 + c0 * s) >> 8) * dim + ((c * 256 + c1 * s) >> 8)
                    x2 = ((r * 256 + c2 * s) >> 8) * dim + ((c * 256 + c3 * s) >> 8)
                else:
                    lim = rows - 1  # quirk Q1: nrows-1 clamps the columns too (pigo.go:168,171)
                    r1 = np.minimum(lim, np.maximum(0, 65536 * r + qcos * c0 - qsin * c1) >> 16)
                    k1 = np.minimum(lim, np.maximum(0, 65536 * c + qsin * c0 + qcos * c1) >> 16)
                    r2 = np.minimum(lim, np.maximum(0, 65536 * r + qcos * c2 - qsin * c3) >> 16)
                    k2 = np.minimum(lim, np.maximum(0, 65536 * c + qsin * c2 + qcos * c3) >> 16)
                    x1 = np.abs(r1) * dim + np.abs(k1)
                    x2 = np.abs(r2) * dim + np.abs(k2)
                if x1.min() < 0 or x2.min() < 0 or x1.max() >= npix or x2.max() >= npix:
                    raise IndexError("pixel index out of range (Go would panic)")
                idx = 2 * idx + (pix[x1] <= pix[x2])
            o = out[alive] + self.preds[t][idx - nleaf]  # float32 + float32
            keep = o > self.thr[t]  # reject when out <= thr (pigo.go:139)
            out[alive] = o
            alive = alive[keep]
        if self.ntrees == 0:
            return np.zeros((0, 4)), nwin
        q = out[alive] - self.thr[self.ntrees - 1]
        pos = q > 0
        alive, q = alive[pos], q[pos]
        res = np.stack([R[alive], Cc[alive], np.full(alive.size, s), q.astype(np.float64)], axis=1)
        return res, nwin

    def run_cascade(self, pixels, rows, cols, dim, min_size, max_size, shift, scale_factor, angle=0.0):
        pix = np.ascontiguousarray(pixels, dtype=np.uint8).ravel()
        dets, nwin = [], 0
        s = int(min_size)
        while s <= max_size:
            step = int(max(shift * float(s), 1.0))
            d, n = self._scan_scale(pix, pix.size, rows, cols, dim, s, step, angle)
            nwin += n
            if len(d):
                dets.append(d)
            s = int(float(s) + max(2.0, float(s) * scale_factor - float(s)))
        if dets:
            d = np.concatenate(dets, axis=0)
        else:
            d = np.zeros((0, 4))
        return d, nwin

    @staticmethod
    def calc_iou(d1, d2):
        r1, c1, s1 = float(d1[0]), float(d1[1]), float(d1[2])
        r2, c2, s2 = float(d2[0]), float(d2[1]), float(d2[2])
        over_r = max(0.0, min(r1 + s1 / 2, r2 + s2 / 2) - max(r1 - s1 / 2, r2 - s2 / 2))
        over_c = max(0.0, min(c1 + s1 / 2, c2 + s2 / 2) - max(c1 - s1 / 2, c2 - s2 / 2))
        return over_r * over_c / (s1 * s1 + s2 * s2 - over_r * over_c)

    @classmethod
    def cluster_detections(cls, dets, iou_threshold):
        """dets: list of (row, col, scale, q[float32-valued]); STABLE sort by q (see module docstring)."""
        d = sorted([(int(a), int(b), int(c), np.float32(q)) for a, b, c, q in dets], key=lambda t: t[3])
        n = len(d)
        assigned = [False] * n
        clusters = []
        for i in range(n):
            if assigned[i]:
                continue
            r = c = s = cnt = 0
            q = np.float32(0.0)
            for j in range(n):
                if cls.calc_iou(d[i], d[j]) > iou_threshold:
                    assigned[j] = True
                    r += d[j][0]
                    c += d[j][1]
                    s += d[j][2]
                    q = np.float32(q + d[j][3])
                    cnt += 1
            if cnt > 0:
                clusters.append((r // cnt, c // cnt, s // cnt, q))
        return d, clusters


def np_rgb_to_grayscale(pix, kind=0):
    """Independent vectorised restatement of RgbToGrayscale (core/grayscale.go:8-23; kind 2: wasm/canvas/canvas.go:179-191).

    kind 0: *image.NRGBA (color.NRGBA.RGBA(): c*0x101*A/0xff), kind 1: *image.RGBA (c*0x101).  float64, left to right.
    """
    pix = np.asarray(pix, dtype=np.uint8)
    h, w = pix.shape[:2]
    ch = pix.astype(np.uint32)
    if kind == 2:
        v = np.float64(0.2126) * ch[..., 0].astype(np.float64) + np.float64(0.7152) * ch[..., 1].astype(np.float64)
        v = v + np.float64(0.0722) * ch[..., 2].astype(np.float64)
        r = np.where(v >= 0, np.floor(v), np.ceil(v))          # math.Round: truncate, then step away from zero at >= .5
        r = np.where(np.abs(v - r) >= 0.5, r + np.sign(v), r)
        return r.astype(np.uint8).reshape(h * w)
    c16 = ch[..., :3] * np.uint32(0x101)
    if kind == 0:
        c16 = (c16 * ch[..., 3:4]) // np.uint32(0xff)
    f = c16.astype(np.float64)
    v = np.float64(0.299) * f[..., 0] + np.float64(0.587) * f[..., 1]
    v = v + np.float64(0.114) * f[..., 2]
    return np.trunc(v / 256).astype(np.uint8).reshape(h * w)


class NpPuploc:
    """Independent restatement of PuplocCascade (core/puploc.go:22-277), vectorised over the perturbations.

    Written from the Go text separately from oracle/pigo_oracle.c; every float32 step is a numpy float32 operation
    (IEEE single, unfused), every integer step int64."""

    Q_COS = np.array([256, 251, 236, 212, 181, 142, 97, 49, 0, -49, -97, -142, -181, -212, -236, -251, -256, -251, -236, -212, -181, -142,
                      -97, -49, 0, 49, 97, 142, 181, 212, 236, 251, 256], dtype=np.float32)  # puploc.go:163
    Q_SIN = np.array([0, 49, 97, 142, 181, 212, 236, 251, 256, 251, 236, 212, 181, 142, 97, 49, 0, -49, -97, -142, -181, -212, -236, -251,
                      -256, -251, -236, -212, -181, -142, -97, -49, 0], dtype=np.float32)    # puploc.go:164

    def __init__(self, packet: bytes):  # UnpackCascade, puploc.go:38-103
        hdr = np.frombuffer(packet[:16], dtype="<u4")
        self.stages, self.trees, self.depth = int(hdr[0]), int(hdr[2]), int(hdr[3])
        self.scales = np.frombuffer(packet[4:8], dtype="<f4")[0]
        D = 1 << self.depth
        n = self.stages * self.trees
        rec = np.frombuffer(packet, dtype=np.uint8, count=n * (12 * D - 4), offset=16).reshape(n, 12 * D - 4)
        self.codes = rec[:, : 4 * D - 4].copy().view(np.int8).reshape(n, D - 1, 4)  # node k of a tree = bytes 4k..4k+3
        self.preds = rec[:, 4 * D - 4:].copy().view("<f4").reshape(n, D, 2)

    def classify(self, r, c, s, pixels, nrows, ncols, dim, flip_v=False, angle=None):
        """classifyRegion (angle None) / classifyRotatedRegion over arrays of starting points -> (r, c, s) float32 arrays"""
        pix = np.asarray(pixels, dtype=np.uint8).ravel()
        r, c, s = (np.array(v, dtype=np.float32).copy() for v in (r, c, s))
        D = 1 << self.depth
        rot = angle is not None
        if rot:
            k = int(32.0 * angle)
            qsin = (s * self.Q_SIN[k]).astype(np.float32).astype(np.int64)  # int(qsin): truncation
            qcos = (s * self.Q_COS[k]).astype(np.float32).astype(np.int64)
        lane = np.arange(r.size)
        for i in range(self.stages):
            dr = np.zeros(r.size, np.float32)
            dc = np.zeros(r.size, np.float32)
            ri, ci = r.astype(np.int64), c.astype(np.int64)                 # int(r), int(c): truncation toward zero
            sr = np.where(s >= 0, np.floor(s.astype(np.float64) + 0.5), -np.floor(-s.astype(np.float64) + 0.5)).astype(np.int64)
            for j in range(self.trees):
                t = i * self.trees + j
                idx = np.zeros(r.size, np.int64)
                for _ in range(self.depth):
                    cd = self.codes[t][idx].astype(np.int64)                # [P, 4]
                    c1code, c2code = cd[:, 1], cd[:, 3]
                    if flip_v:  # negation happens in int8 and wraps: -(-128) == -128
                        c1code = (-self.codes[t][idx][:, 1]).astype(np.int8).astype(np.int64)
                        c2code = (-self.codes[t][idx][:, 3]).astype(np.int8).astype(np.int64)
                    if rot:
                        r1 = np.minimum(nrows - 1, np.maximum(0, 65536 * ri + qcos * cd[:, 0] - qsin * c1code) >> 16)
                        c1 = np.minimum(ncols - 1, np.maximum(0, 65536 * ci + qsin * cd[:, 0] + qcos * c1code) >> 16)
                        r2 = np.minimum(nrows - 1, np.maximum(0, 65536 * ri + qcos * cd[:, 2] - qsin * c2code) >> 16)
                        c2 = np.minimum(ncols - 1, np.maximum(0, 65536 * ci + qsin * cd[:, 2] + qcos * c2code) >> 16)
                        bit = pix[r1 * dim + c1] <= pix[r2 * dim + c2]
                    else:
                        r1 = np.minimum(nrows - 1, np.maximum(0, (256 * ri + cd[:, 0] * sr) >> 8))
                        r2 = np.minimum(nrows - 1, np.maximum(0, (256 * ri + cd[:, 2] * sr) >> 8))
                        c1 = np.minimum(ncols - 1, np.maximum(0, (256 * ci + c1code * sr) >> 8))
                        c2 = np.minimum(ncols - 1, np.maximum(0, (256 * ci + c2code * sr) >> 8))
                        bit = pix[r1 * dim + c1] > pix[r2 * dim + c2]
                    idx = 2 * idx + 1 + bit.astype(np.int64)
                leaf = self.preds[t][idx - (D - 1)]                          # [P, 2]
                dr = (dr + leaf[:, 0]).astype(np.float32)
                dc = (dc + (-leaf[:, 1] if flip_v else leaf[:, 1])).astype(np.float32)
            r = (r + (dr * s).astype(np.float32)).astype(np.float32)
            c = (c + (dc * s).astype(np.float32)).astype(np.float32)
            s = (s * self.scales).astype(np.float32)
        del lane
        return r, c, s

    def run_detector(self, row, col, scale, perturbs, pixels, rows, cols, dim, angle, flip_v, rnd, pool=None):
        """RunDetector, puploc.go:239-277 (pool: float32 [3, 63] in/out or None for a new sync.Pool object)"""
        det = np.zeros((3, 63), np.float32) if pool is None else pool
        u = np.asarray(rnd, dtype=np.float32).reshape(-1)[: 3 * perturbs].reshape(perturbs, 3)
        f = np.float32
        sc0 = f(scale)
        rr = (f(row) + ((sc0 * f(0.15)) * (f(0.5) - u[:, 0])).astype(np.float32)).astype(np.float32)
        cc = (f(col) + ((sc0 * f(0.15)) * (f(0.5) - u[:, 1])).astype(np.float32)).astype(np.float32)
        ss = (sc0 * (f(0.925) + (f(0.15) * u[:, 2]).astype(np.float32)).astype(np.float32)).astype(np.float32)
        a = None if not angle > 0.0 else min(angle, 1.0)
        r, c, s = self.classify(rr, cc, ss, pixels, rows, cols, dim, flip_v, a)
        det[0, :perturbs], det[1, :perturbs], det[2, :perturbs] = r, c, s
        det.sort(axis=1)
        mid = int(np.floor(perturbs / 2 + 0.5))
        return int(det[0, mid]), int(det[1, mid]), np.float32(det[2, mid])
