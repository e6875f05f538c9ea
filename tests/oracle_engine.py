"""An oracle-backed stand-in for pigo_amd.pipeline.HipEngine (TEST INFRASTRUCTURE): the same five compute steps served
by the CPU oracle, so that the host logic of pigo_amd/pipeline.py can be checked on CPU and the HIP engine's output can
be compared document for document."""
import numpy as np

import oracle
from pigo_amd import core


class OracleEngine:
    def unpack(self, packet):
        return oracle.OraclePigo.unpack(packet)

    def unpack_puploc(self, packet):
        return oracle.OraclePuploc.unpack(packet)

    def rgb_to_grayscale(self, rgba, kind):
        return oracle.rgb_to_grayscale(rgba, kind)

    def run_cascade(self, classifier, pixels, rows, cols, dim, mn, mx, shift, scale, angle):
        return classifier.run_cascade(pixels, rows, cols, dim, mn, mx, shift, scale, angle)

    def cluster_detections(self, classifier, dets, iou):
        return classifier.cluster_detections(dets, iou)

    def run_detector(self, plc, pl, pixels, rows, cols, dim, angle, flip_v, rnd, pool):
        r, c, s = plc.run_detector(pl.Row, pl.Col, pl.Scale, pl.Perturbs, pixels, rows, cols, dim, angle, flip_v, rnd, pool)
        return core.Puploc(r, c, float(s), 0)

    def get_landmark_point(self, flpc, left, right, pixels, rows, cols, dim, perturb, flip_v, rnd, pool):
        r, c, s = flpc.get_landmark_point((left.Row, left.Col, np.float32(left.Scale)), (right.Row, right.Col, np.float32(right.Scale)), pixels,
                                          rows, cols, dim, perturb, flip_v, rnd, pool)
        return core.Puploc(r, c, float(s), 0)
