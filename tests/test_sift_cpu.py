"""SURVEY section 8f row 4 on the CPU side: Rotation::getRotatedWindowPoints, rotateImageSection + SIFT
(Diff_of_Gauss.cpp:528-559,561-693, rotation.cpp:112-130) in the oracle against analytic answers and an
independent whole-array numpy restatement (tests/npref.py), on the reference's own images."""
import ctypes
import ctypes.util

import numpy as np
import pytest

import oracle
from tests import npref, refimg


def test_cos_sin_float_and_double_forms_agree_for_every_histogram_angle():
    # the reference's unqualified cos(angle) (rotation.cpp:16) is ::cos(double) or a float overload
    # depending on OpenCV's includes; the pipeline only produces multiples of 10 degrees
    # (Diff_of_Gauss.cpp:363) and for those both forms give the same float with glibc
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.cosf.restype = libm.sinf.restype = ctypes.c_float
    libm.cosf.argtypes = libm.sinf.argtypes = [ctypes.c_float]
    for a in range(0, 360, 10):
        c, s = oracle.cos_sin_deg(a)
        ang = np.float32(np.float64(np.float32(a)) * (np.float64(3.1415926535897932384626433832795) / np.float64(np.float32(180.0))))
        assert np.float32(libm.cosf(float(ang))) == c and np.float32(libm.sinf(float(ang))) == s, a
    assert oracle.cos_sin_deg(0) == (1.0, 0.0)


def test_rotated_window_points_known_answers():
    p0 = oracle.rotated_window_points(100, 50, 16, 0.0)
    jj, ii = np.meshgrid(np.arange(92, 109), np.arange(42, 59))
    assert p0.shape == (289, 2) and (p0[:, 0] == jj.ravel()).all() and (p0[:, 1] == ii.ravel()).all()
    # 90 degrees clockwise in image coordinates: (dx, dy) -> (-dy, dx) up to the float cosine of pi/2
    # (cos = -4.37e-8: products of magnitude < 1 truncate towards zero)
    p90 = oracle.rotated_window_points(100, 50, 16, 90.0)
    dx, dy = (jj - 100).ravel(), (ii - 50).ravel()
    c, s = oracle.cos_sin_deg(90.0)
    assert s == 1.0 and abs(c) < 1e-7
    wx = np.trunc((dx.astype(np.float32) * c) - (dy.astype(np.float32) * s)).astype(int) + 100
    wy = np.trunc((dx.astype(np.float32) * s) + (dy.astype(np.float32) * c)).astype(int) + 50
    assert (p90[:, 0] == wx).all() and (p90[:, 1] == wy).all()
    for cx, cy, w, th in [(30, 40, 16, 10.0), (500, 20, 16, 350.0), (64, 64, 8, 45.0), (20, 20, 16, 180.0)]:
        assert (oracle.rotated_window_points(cx, cy, w, th) == npref.rotated_window_points(cx, cy, w, th)).all()
    # integer truncation collapses neighbours: a rotated window is not a bijection (rotation.cpp:22-23)
    assert len({tuple(p) for p in oracle.rotated_window_points(100, 50, 16, 30.0)}) < 289


@pytest.mark.parametrize("name", ["blox", "home"])
def test_sift_descriptors_match_the_numpy_restatement(name):
    img = refimg.load(name)
    p = oracle.Pyramid(img, 4, 1.6)
    n_def = n_undef = 0
    for o in range(4):
        oriented = p.filter_keypoints(o, p.keypoints(o, 3))
        desc, ok = p.sift_descriptors(o, oriented)
        assert desc.shape == (len(oriented), 128)
        for q in range(min(len(oriented), 40)):
            kp = oriented[q]
            sigma = 1.5 * p.sigmas[o][kp["level"]]
            k = oracle.gauss_kernel_f32(oracle.gauss_ksize_f32(sigma), sigma)
            want = npref.sift_descriptor(p.gauss(o, int(kp["level"])), int(kp["row"]), int(kp["col"]), float(kp["value"]), k)
            if want is None:
                assert not ok[q] and not desc[q].any()
                n_undef += 1
            else:
                assert ok[q]
                assert np.array_equal(desc[q], want, equal_nan=True), (o, q)
                n_def += 1
                fin = desc[q][np.isfinite(desc[q])]
                if len(fin) == 128:  # second normalisation: the clipped maximum becomes exactly 1
                    assert fin.max() == 1.0 and fin.min() >= 0.0
    p.close()
    assert n_def > 20
    if name == "home":  # landscape: Mat::at<>(x, y) with x as the row runs off the padded level for large columns
        assert n_undef > 0
    else:               # square image: every keypoint's window stays inside
        assert n_undef == 0


def test_descriptor_of_a_flat_window_is_nan_like_the_reference():
    # all-zero magnitudes: 0 / 0 in the first normalisation (Diff_of_Gauss.cpp:661), NaN to the end
    img = np.full((64, 64), 90, np.uint8)
    img[:, 40:] = 91  # one faint edge far from the sampled window
    p = oracle.Pyramid(img, 1, 1.6)
    kp = np.zeros(1, oracle.POINT_DTYPE)
    kp["row"], kp["col"], kp["value"], kp["level"] = 10, 10, 0, 1
    desc, ok = p.sift_descriptors(0, kp)
    p.close()
    assert ok[0] and np.isnan(desc[0]).all()


def test_c_abi_host_helpers_match_the_oracle(tmp_path):
    # vslam_cos_sin_deg / vslam_rotated_window_points / vslam_descriptor_file_write are pure host
    # computations of the product library (no GPU needed)
    from visualslam_amd import capi

    capi.build()
    for a in list(range(0, 360, 10)) + [12.5, -30.0, 721.0]:
        assert capi.cos_sin_deg(a) == oracle.cos_sin_deg(a)
    rng = np.random.default_rng(3)
    for _ in range(50):
        cx, cy = int(rng.integers(-50, 4000)), int(rng.integers(-50, 4000))
        w = int(rng.choice([2, 8, 16, 17]))
        th = float(rng.choice(np.arange(0, 360, 10)))
        assert (capi.rotated_window_points(cx, cy, w, th) == oracle.rotated_window_points(cx, cy, w, th)).all()
    d = rng.random((5, 128), dtype=np.float32)
    d[2, 7] = np.nan
    path = tmp_path / "featureDescriptors.dat"
    capi.descriptor_file_write(str(path), d)
    raw = open(path, "rb").read()
    assert np.frombuffer(raw[:12], "<i4").tolist() == [5, 128, 24]  # Diff_of_Gauss.cpp:842-849
    assert raw[12:] == d.tobytes()
    capi.descriptor_file_write(str(path), np.zeros((0, 128), np.float32))
    assert open(path, "rb").read() == np.array([0, 128, 24], "<i4").tobytes()
