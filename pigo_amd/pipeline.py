"""Result plumbing of the reference's consumers on top of the HIP path.

Two consumers of the hot path exist in the reference tree and both are reproduced here as thin HOST code over
``pigo_amd.core`` (every pixel-touching step -- RgbToGrayscale, RunCascade, ClusterDetections, RunDetector,
GetLandmarkPoint -- runs on the GPU through the C ABI; there is no CPU path behind this module):

* the CLI's detection flow and JSON output, cmd/pigo/main.go:236-348 (detectFaces), :358-578 (drawFaces, minus the
  drawing) and :88-100,224 (the ``coord`` / ``detection`` schema, written with ``json.NewEncoder(out).Encode``).
  The schema's quirks are kept because downstream tools parse them: ``coord.Row`` is tagged "x" and ``coord.Col`` "y"
  and the CLI stores columns in ``Row`` and rows in ``Col`` (main.go:394-398); every int field is ``omitempty``
  (a zero coordinate disappears); ``eyesCoords`` / ``landmarkCoords`` are declared outside the per-face loop
  (main.go:363-366), so face k's lists also hold the points of faces 0..k-1.
* the c-shared ``FindFaces`` entry of examples/facedet/pigo.go:22-57 with its Python caller
  (examples/facedet/demo.py:14-41), replaced by ``find_faces``: same parameters (100..600, shift 0.15, scale 1.1,
  iou 0, Q >= 5.0), results as a plain (n, 3) int array instead of a pointer into the Go heap.
"""
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np

from . import core

EYE_CASCADES = ["lp46", "lp44", "lp42", "lp38", "lp312"]   # cmd/pigo/main.go:69
MOUTH_CASCADES = ["lp93", "lp84", "lp82", "lp81"]          # cmd/pigo/main.go:70
PERTURB = 63                                                # cmd/pigo/main.go:54
Q_THRESH = np.float32(5.0)                                  # cmd/pigo/main.go:360


class DetectorState:
    """The process-global state RunDetector reads in the reference: the math/rand stream and the sync.Pool object."""

    def __init__(self, float32_stream: Optional[Callable[[int], np.ndarray]] = None, use_pool: bool = True):
        rng = np.random.default_rng()
        self.draw = float32_stream if float32_stream is not None else (lambda n: rng.random(n, dtype=np.float32))
        self.pool = core.new_pool() if use_pool else None


@dataclass
class Coord:
    """type coord, cmd/pigo/main.go:88-92 (JSON: Row -> "x", Col -> "y", Scale -> "size", all omitempty)"""
    Row: int = 0
    Col: int = 0
    Scale: int = 0


@dataclass
class DetectionRecord:
    """type detection, cmd/pigo/main.go:95-99"""
    EyePoints: List[Coord] = field(default_factory=list)
    LandmarkPoints: List[Coord] = field(default_factory=list)
    FacePoints: Coord = field(default_factory=Coord)


def _coord_json(c: Coord) -> str:
    parts = []
    if c.Row != 0:
        parts.append('"x":%d' % c.Row)
    if c.Col != 0:
        parts.append('"y":%d' % c.Col)
    if c.Scale != 0:
        parts.append('"size":%d' % c.Scale)
    return "{" + ",".join(parts) + "}"


def encode_json(dets: List[DetectionRecord]) -> str:
    """json.NewEncoder(out).Encode(dets) (cmd/pigo/main.go:224): compact, struct field order, omitempty, trailing newline."""
    items = []
    for d in dets:
        parts = []
        if d.EyePoints:
            parts.append('"eyes":[' + ",".join(_coord_json(c) for c in d.EyePoints) + "]")
        if d.LandmarkPoints:
            parts.append('"landmark_points":[' + ",".join(_coord_json(c) for c in d.LandmarkPoints) + "]")
        parts.append('"face":' + _coord_json(d.FacePoints))  # a struct is never "empty" for omitempty
        items.append("{" + ",".join(parts) + "}")
    return "[" + ",".join(items) + "]\n"


class HipEngine:
    """The compute steps, all on the GPU through pigo_amd.core (tests swap in an oracle-backed engine to check the host
    logic of this module; the product never does)."""

    def __init__(self, device: int = 0):
        self.device = device

    def unpack(self, packet: bytes):
        return core.NewPigo(self.device).Unpack(packet)

    def unpack_puploc(self, packet: bytes):
        return core.NewPuplocCascade(self.device).UnpackCascade(packet)

    def rgb_to_grayscale(self, rgba: np.ndarray, kind: int) -> np.ndarray:
        return core.RgbToGrayscale(rgba, kind=kind, device=self.device)

    def run_cascade(self, classifier, pixels, rows, cols, dim, mn, mx, shift, scale, angle):
        cp = core.CascadeParams(MinSize=mn, MaxSize=mx, ShiftFactor=shift, ScaleFactor=scale,
                                ImageParams=core.ImageParams(Pixels=pixels, Rows=rows, Cols=cols, Dim=dim))
        return classifier.RunCascade(cp, angle)

    def cluster_detections(self, classifier, dets, iou):
        return classifier.ClusterDetections(dets, iou)

    def run_detector(self, plc, pl: "core.Puploc", pixels, rows, cols, dim, angle, flip_v, rnd, pool) -> "core.Puploc":
        return plc.RunDetector(pl, core.ImageParams(Pixels=pixels, Rows=rows, Cols=cols, Dim=dim), angle, flip_v, rnd=rnd, pool=pool)

    def get_landmark_point(self, flpc, left, right, pixels, rows, cols, dim, perturb, flip_v, rnd, pool) -> "core.Puploc":
        return flpc.GetLandmarkPoint(left, right, core.ImageParams(Pixels=pixels, Rows=rows, Cols=cols, Dim=dim), perturb, flip_v, rnd=rnd, pool=pool)


@dataclass
class FaceDetector:
    """type faceDetector, cmd/pigo/main.go:73-86, with the CLI's flag defaults (:105-118).  Cascades are passed as bytes
    (the CLI reads them from -cf / -plc / -flpc)."""
    cascade: bytes = b""
    puploc: bytes = b""
    flploc: Dict[str, bytes] = field(default_factory=dict)  # file name -> cascade, what ReadCascadeDir returns (flploc.go:60-81)
    angle: float = 0.0
    minSize: int = 20
    maxSize: int = 1000
    shiftFactor: float = 0.15
    scaleFactor: float = 1.15
    iouThreshold: float = 0.15
    engine: object = None
    state: DetectorState = None

    def __post_init__(self):
        self.engine = self.engine if self.engine is not None else HipEngine()
        self.state = self.state if self.state is not None else DetectorState()
        self._classifier = self._plc = None
        self._flpcs = {}
        self._img = None

    @staticmethod
    def read_cascade_dir(path: str) -> Dict[str, bytes]:
        """The files ReadCascadeDir (core/flploc.go:60-81) would unpack, by name."""
        names = sorted(os.listdir(path))
        if not names:
            raise core.PigoError("the provided directory is empty")
        return {n: open(os.path.join(path, n), "rb").read() for n in names}

    # detectFaces, cmd/pigo/main.go:236-348 (image decoding is the caller's: pass the decoded NRGBA pixels or a gray frame)
    def detect_faces(self, src: np.ndarray, kind: int = core.PIX_NRGBA) -> np.ndarray:
        src = np.asarray(src)
        if src.ndim == 3:
            rows, cols = src.shape[:2]                                     # src.Bounds().Max.Y / .X   (:275)
            pixels = self.engine.rgb_to_grayscale(src, kind)               # pigo.RgbToGrayscale(src)  (:274)
        else:
            rows, cols = src.shape
            pixels = np.ascontiguousarray(src, dtype=np.uint8).ravel()
        self._img = (pixels, rows, cols, cols)                             # imgParams, Dim = cols     (:280-285)
        if self._classifier is None:
            self._classifier = self.engine.unpack(self.cascade)            # p.Unpack(cascadeFile)     (:311)
        if self.puploc and self._plc is None:
            self._plc = self.engine.unpack_puploc(self.puploc)             # plcReader                 (:316-334)
        if self.flploc and not self._flpcs:
            if not self.puploc:
                raise core.PigoError("the puploc cascade file is required: use the -plc flag")  # :337-340
            self._flpcs = {n: self.engine.unpack_puploc(b) for n, b in self.flploc.items()}
        faces = self.engine.run_cascade(self._classifier, pixels, rows, cols, cols, self.minSize, self.maxSize, self.shiftFactor,
                                        self.scaleFactor, self.angle)                            # :348
        return self.engine.cluster_detections(self._classifier, faces, self.iouThreshold)        # :351

    def _run(self, plc, row, col, scale, flip_v=False):
        pix, rows, cols, dim = self._img
        return self.engine.run_detector(plc, core.Puploc(row, col, scale, PERTURB), pix, rows, cols, dim, self.angle, flip_v,
                                        self.state.draw(3 * PERTURB), self.state.pool)

    def _flp(self, name, left, right, flip_v):
        pix, rows, cols, dim = self._img
        return self.engine.get_landmark_point(self._flpcs[name], left, right, pix, rows, cols, dim, PERTURB, flip_v,
                                              self.state.draw(3 * PERTURB), self.state.pool)

    # drawFaces, cmd/pigo/main.go:358-578, without the drawing
    def draw_faces(self, faces: np.ndarray) -> List[DetectionRecord]:
        detections: List[DetectionRecord] = []
        eyes: List[Coord] = []        # declared outside the loop in the reference (:363-364): cumulative over faces
        landmarks: List[Coord] = []   # (:365)
        f32 = np.float32
        for face in faces:
            if not f32(face["q"]) > Q_THRESH:                                                     # :370
                continue
            row, col, scale = int(face["row"]), int(face["col"]), int(face["scale"])
            face_coord = Coord(Col=row - scale // 2, Row=col - scale // 2, Scale=scale)           # :394-398 (sic: swapped)
            if self.puploc and scale > 50:                                                        # :404
                lrow = row - int(f32(0.075) * f32(scale))                                         # :417-421
                left = self._run(self._plc, lrow, col - int(f32(0.175) * f32(scale)), float(f32(scale) * f32(0.25)))
                if left.Row > 0 and left.Col > 0:                                                 # :423
                    eyes.append(Coord(Col=left.Row, Row=left.Col, Scale=int(f32(left.Scale))))    # :446-450
                right = self._run(self._plc, lrow, col + int(f32(0.185) * f32(scale)), float(f32(scale) * f32(0.25)))  # :454-461
                if right.Row > 0 and right.Col > 0:
                    eyes.append(Coord(Col=right.Row, Row=right.Col, Scale=int(f32(right.Scale))))  # :484-488
                if self.flploc:                                                                   # :491
                    seq = [(n, fl) for n in EYE_CASCADES for fl in (False, True)] + [(n, False) for n in MOUTH_CASCADES] + [("lp84", True)]
                    for name, flip in seq:                                                        # :492-563
                        flp = self._flp(name, left, right, flip)
                        if flp.Row > 0 and flp.Col > 0:
                            landmarks.append(Coord(Col=flp.Row, Row=flp.Col, Scale=int(f32(flp.Scale))))
            detections.append(DetectionRecord(FacePoints=face_coord, EyePoints=list(eyes), LandmarkPoints=list(landmarks)))  # :566-570
        return detections

    def detect_json(self, src: np.ndarray, kind: int = core.PIX_NRGBA) -> str:
        """`pigo -in src -json -`: the JSON document the CLI writes, or "" when no face was found (main.go:215-232)."""
        dets = self.draw_faces(self.detect_faces(src, kind))
        return encode_json(dets) if dets else ""


def find_faces(pixels: np.ndarray, rows: int = 480, cols: int = 640, cascade: bytes = None, classifier=None, engine=None) -> np.ndarray:
    """FindFaces of examples/facedet/pigo.go:22-57 / clusterDetection :61-101 as a plain function: gray frame in,
    (n, 3) int64 array of (row, col, scale) for the clusters with Q >= 5.0 out."""
    engine = engine if engine is not None else HipEngine()
    if classifier is None:
        if cascade is None:
            from . import synth
            cascade = synth.facefinder_bytes()
        classifier = engine.unpack(cascade)
    pix = np.ascontiguousarray(pixels, dtype=np.uint8).ravel()
    dets = engine.run_cascade(classifier, pix, rows, cols, cols, 100, 600, 0.15, 1.1, 0.0)     # pigo.go:62-74,97
    dets = engine.cluster_detections(classifier, dets, 0.0)                                    # pigo.go:100
    keep = [(int(d["row"]), int(d["col"]), int(d["scale"])) for d in dets if np.float32(d["q"]) >= np.float32(5.0)]  # pigo.go:29-33
    return np.array(keep, dtype=np.int64).reshape(-1, 3)
