"""Loader for the reference-held test images committed as raw gray fixtures
(tests/golden/ref_images/*.pgm.gz, written by tests/golden/make_ref_images.py from
/root/reference/KeyPointDetection/images -- data only, never read from /root/reference at test time)."""
import gzip
import hashlib
import json
import os

import numpy as np

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_images")
NAMES = ("blox", "home", "building", "chessboard")
MANIFEST = json.load(open(os.path.join(DIR, "manifest.json")))
# octave sizes (rows, cols) of GaussPyramid(img, 4, 1.6) and the automatic
# octave count of the second constructor (GaussPyramid.cpp:150-152)
OCTAVE_SIZES = {
    "blox": [(512, 512), (256, 256), (128, 128), (64, 64)],
    "home": [(768, 1024), (384, 512), (192, 256), (96, 128)],
    "building": [(1200, 1736), (600, 868), (300, 434), (150, 217)],
    "chessboard": [(2480, 3508), (1240, 1754), (620, 877), (310, 438)],
}
AUTO_OCTAVES = {"blox": 4, "home": 4, "building": 5, "chessboard": 6}


def load(name: str) -> np.ndarray:
    raw = gzip.decompress(open(os.path.join(DIR, name + ".pgm.gz"), "rb").read())
    magic, dims, maxv, data = raw.split(b"\n", 3)
    assert magic == b"P5" and maxv == b"255"
    cols, rows = map(int, dims.split())
    img = np.frombuffer(data, np.uint8).reshape(rows, cols).copy()
    m = MANIFEST[name]
    assert (rows, cols) == (m["rows"], m["cols"])
    assert hashlib.sha256(img.tobytes()).hexdigest() == m["gray_sha256"], "fixture bytes changed"
    return img
