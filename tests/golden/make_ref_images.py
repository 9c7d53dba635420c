#!/usr/bin/env python3
"""Converts the reference's four sample images to raw 8-bit gray PGM fixtures.

    python tests/golden/make_ref_images.py        (build container only: needs /root/reference + Pillow)

The images under /root/reference/KeyPointDetection/images are the only test inputs the reference
holds (Harris_corners.cpp:148 chessboard.png, Diff_of_Gauss.cpp:730 home.jpg,
tests/GaussPyramid_Test.cpp:78 building.jpg, tests/rotate_image_test.cpp:25 blox.jpg).  They are
DATA; what is committed is their decoded gray raster (binary PGM "P5", gzip), nothing else of
the reference.  Decoding: JPEGs through libjpeg's own grayscale output (Pillow draft mode 'L' --
the luma plane, which is also what cv::imread(IMREAD_GRAYSCALE) hands back for a JPEG); the
RGBA PNG has R == G == B and alpha == 255 everywhere (asserted), so its gray raster is the R plane.
Parity is defined from these raw gray bytes onward: image decoding is
off the hot path and decoders differ by +-1.
"""
import gzip
import hashlib
import json
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/KeyPointDetection/images"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_images")


def gray(path):
    im = Image.open(path)
    if im.format == "JPEG":
        im.draft("L", im.size)
        im = im.convert("L")
        return np.asarray(im, dtype=np.uint8)
    a = np.asarray(im)
    assert a.ndim == 3 and a.shape[2] == 4 and (a[..., 3] == 255).all()
    assert (a[..., 0] == a[..., 1]).all() and (a[..., 0] == a[..., 2]).all()
    return np.ascontiguousarray(a[..., 0])


def main():
    os.makedirs(DST, exist_ok=True)
    manifest = {}
    for name in ("blox.jpg", "home.jpg", "building.jpg", "chessboard.png"):
        g = gray(os.path.join(SRC, name))
        stem = os.path.splitext(name)[0]
        pgm = b"P5\n%d %d\n255\n" % (g.shape[1], g.shape[0]) + g.tobytes()
        with open(os.path.join(DST, stem + ".pgm.gz"), "wb") as f:
            f.write(gzip.compress(pgm, 9, mtime=0))
        manifest[stem] = {
            "source": "KeyPointDetection/images/" + name,
            "source_sha256": hashlib.sha256(open(os.path.join(SRC, name), "rb").read()).hexdigest(),
            "rows": int(g.shape[0]), "cols": int(g.shape[1]),
            "gray_sha256": hashlib.sha256(g.tobytes()).hexdigest(),
        }
        print(stem, g.shape, manifest[stem]["gray_sha256"][:16])
    with open(os.path.join(DST, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
