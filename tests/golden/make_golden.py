"""Generate the committed golden vectors from the CPU oracle.

The reference has no golden data and cannot be built here (no OpenCV), so these vectors are
outputs of oracle/vslam_oracle.c on seeded synthetic inputs.
They pin the oracle against accidental change and give the GPU parity tests fixed targets.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from visualslam_amd import synth  # noqa: E402

CASES = {
    # name: (rows, cols, kind, stream_id, octaves)
    "checker_48x64": (48, 64, "checker", 1, 3),
    "noise_40x56": (40, 56, "noise", 2, 3),
    "checker_33x47": (33, 47, "checker", 3, 2),  # odd sizes: ragged tiles, half-even size chain
}


def make(name):
    rows, cols, kind, sid, n_oct = CASES[name]
    img = synth.frame_np(rows, cols, 0, sid, kind)
    d = {"img": img}
    R = oracle.harris_response(img)
    d["response"] = R
    d["nms_mask"] = oracle.nms_strict(oracle.convert_scale_abs(R), 3)
    n2, tmax = oracle.nms2(R, 5)
    d["nms2"] = n2
    d["nms2_true_max"] = np.float32(tmax)
    d["harris_kps"] = oracle.harris_keypoints(n2)
    p = oracle.Pyramid(img, n_oct, 1.6)
    d["n_octaves"] = np.int32(n_oct)
    for o in range(n_oct):
        d[f"base_{o}"] = p.base(o)
        d[f"gauss_{o}"] = np.stack([p.gauss(o, l) for l in range(6)])
        d[f"dog_{o}"] = np.stack([p.dog(o, l) for l in range(5)])
        m, pts = p.extrema(o, 3, 8)
        d[f"ext_mask_{o}"] = m
        d[f"ext_pts_{o}"] = pts
        d[f"kp_pts_{o}"] = p.keypoints(o, 3)  # through FeaturePointLocalization (section 8f row 2)
        d[f"oriented_pts_{o}"] = p.filter_keypoints(o, d[f"kp_pts_{o}"])  # filterKeypoints (row 3)
        desc, ok = p.sift_descriptors(o, d[f"oriented_pts_{o}"])            # SIFT descriptors (row 4)
        d[f"sift_desc_{o}"] = np.nan_to_num(desc, nan=-1.0)                   # all-NaN rows (flat windows) stored as -1
        d[f"sift_defined_{o}"] = ok
    return d


def make_opencv(name):
    """The same vectors with every plane that is the output of OpenCV calls alone MADE BY A REAL OpenCV (cv2), in the reference's
    call order (Harris_corners.cpp:158-164, GaussPyramid.cpp:110,126,177,197), and an `opencv_version` stamp.  The fields the
    reference's own loops produce (response, NMS maps, lattice extrema, keypoint lists) are the oracle's on those planes; the
    call is refused when any OpenCV-made plane differs from the oracle's - then the recalled rule is wrong for this OpenCV and
    tests/test_opencv_crosscheck.py says which (tools/pin_with_opencv.sh runs it first)."""
    import cv2

    d = make(name)
    img = d["img"]
    b = cv2.GaussianBlur(img, (3, 3), 0, borderType=cv2.BORDER_DEFAULT)
    ix = cv2.Sobel(b, cv2.CV_32F, 1, 0, ksize=1, scale=1, delta=0, borderType=cv2.BORDER_DEFAULT)
    iy = cv2.Sobel(b, cv2.CV_32F, 0, 1, ksize=1, scale=1, delta=0, borderType=cv2.BORDER_DEFAULT)
    assert (oracle.harris_from_grad(ix, iy) == d["response"]).all(), "Harris planes differ from the oracle's"
    base = cv2.resize(img, None, fx=2, fy=2, interpolation=cv2.INTER_LINEAR)
    for o in range(int(d["n_octaves"])):
        g = [cv2.GaussianBlur(base, (0, 0), oracle.sigma_at(1.6, o, l), sigmaY=0, borderType=cv2.BORDER_DEFAULT) for l in range(6)]
        dg = [cv2.subtract(g[l + 1], g[l]) for l in range(5)]
        for key, planes in ((f"base_{o}", base), (f"gauss_{o}", np.stack(g)), (f"dog_{o}", np.stack(dg))):
            assert (planes == d[key]).all(), f"{key} from OpenCV {cv2.__version__} differs from the oracle's"
            d[key] = planes
        base = cv2.resize(g[3], None, fx=0.5, fy=0.5, interpolation=cv2.INTER_NEAREST)
    d["opencv_version"] = np.array(cv2.__version__)
    return d


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    maker = make_opencv if "--opencv" in sys.argv[1:] else make
    for name in CASES:
        np.savez_compressed(os.path.join(here, name + ".npz"), **maker(name))
        print("wrote", name, "(planes from a real OpenCV)" if maker is make_opencv else "")
