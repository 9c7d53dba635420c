"""Cross-check of the CPU oracle against a REAL OpenCV, call for call.

OpenCV is absent from the build image and from the GPU box, so this module skips itself there
(`pytest.importorskip("cv2")`) -- it exists so that the pin closes the day an OpenCV appears: every
test issues the same OpenCV call the reference makes (file:line in the test) on the reference's own
images and compares the result with the oracle function that restates it.  A failure here means the
ORACLE's recalled rule is wrong for that OpenCV build (the rules marked with a double dagger in
SURVEY.md Appendix A), not that the GPU path is: the fix then belongs in the one function that
isolates the rule, on both sides.  Not exercised in this repository's CI; informational by design.
"""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

import oracle  # noqa: E402
from tests import refimg  # noqa: E402

IMAGES = ["blox", "home"]


@pytest.fixture(scope="module", params=IMAGES)
def img(request):
    return refimg.load(request.param)


def test_harris_pre_blur(img):
    # GaussianBlur(img, blurred, Size(3,3), 0, 0, BORDER_DEFAULT), Harris_corners.cpp:158
    assert (cv2.GaussianBlur(img, (3, 3), 0, borderType=cv2.BORDER_DEFAULT) == oracle.gaussian_blur_u8(img, 3, 0.0)).all()


def test_sobel_ksize_1(img):
    # Sobel(blurred, grad, CV_32F, 1, 0, ksize = 1, 1, 0, BORDER_DEFAULT), Harris_corners.cpp:163-164
    b = oracle.gaussian_blur_u8(img, 3, 0.0)
    assert (cv2.Sobel(b, cv2.CV_32F, 1, 0, ksize=1, scale=1, delta=0, borderType=cv2.BORDER_DEFAULT) == oracle.sobel_k1(b, 1, 0)).all()
    assert (cv2.Sobel(b, cv2.CV_32F, 0, 1, ksize=1, scale=1, delta=0, borderType=cv2.BORDER_DEFAULT) == oracle.sobel_k1(b, 0, 1)).all()


def test_convert_scale_abs():
    # convertScaleAbs(HResponse, abs_HResponse), Harris_corners.cpp:176: rounding, saturation and the
    # x86 cvRound wrap from 2^31 on (oracle/vslam_oracle.c: cvt_abs_u8)
    x = np.array([[0.5, 1.5, 2.5, -3.5, 253.5, 254.5, 255.4, 255.5, 1e9, 2147483520.0, 2147483648.0, 1e12, -1e12]], np.float32)
    x = np.tile(x, (3, 7))  # long enough rows for the SIMD path and the scalar tail
    assert (cv2.convertScaleAbs(x) == oracle.convert_scale_abs(x)).all()


def test_non_maximum_suppression_dilate(img):
    # NonMaximumSuppression, Harris_corners.cpp:70-81: dilate with a 3x3 ones kernel, centre 0
    R = oracle.convert_scale_abs(oracle.harris_response(img))
    k = np.ones((3, 3), np.uint8)
    k[1, 1] = 0
    want = ((R > cv2.dilate(R, k)) * 255).astype(np.uint8)
    assert (want == oracle.nms_strict(R, 3)).all()


def test_pyramid_resizes(img):
    # resize(img, pyrBase, Size(), 2, 2, INTER_LINEAR), GaussPyramid.cpp:110
    assert (cv2.resize(img, None, fx=2, fy=2, interpolation=cv2.INTER_LINEAR) == oracle.resize_linear2x(img)).all()
    # resize(gaussians[3], pyrBase, Size(), 0.5, 0.5, INTER_NEAREST), GaussPyramid.cpp:126 (odd sizes included)
    for a in (img, img[:-1, :-3]):
        assert (cv2.resize(a, None, fx=0.5, fy=0.5, interpolation=cv2.INTER_NEAREST) == oracle.resize_nearest_half(np.ascontiguousarray(a))).all()


def test_pyramid_blurs_and_dog(img):
    # GaussianBlur(img, blurred, Size(0,0), sigma, 0, BORDER_DEFAULT) on CV_8U, GaussPyramid.cpp:177, for
    # every sigma of a 4-octave pyramid (kernel widths 11 ... 245, 8.8 fixed-point taps), and the
    # saturating CV_8U subtraction of :197
    base = oracle.resize_linear2x(img)
    p = oracle.Pyramid(img, 4, 1.6)
    for o in range(4):
        assert (p.base(o) == base).all()
        g = [cv2.GaussianBlur(base, (0, 0), p.sigmas[o][l], sigmaY=0, borderType=cv2.BORDER_DEFAULT) for l in range(6)]
        for l in range(6):
            assert (g[l] == p.gauss(o, l)).all(), ("gauss", o, l, p.sigmas[o][l])
        for l in range(5):
            assert (cv2.subtract(g[l + 1], g[l]) == p.dog(o, l)).all(), ("dog", o, l)
        base = cv2.resize(g[3], None, fx=0.5, fy=0.5, interpolation=cv2.INTER_NEAREST)
    p.close()


# the f32 stages exist in the oracle in two variants - every multiply and add rounded (OpenCV's SSE2
# baseline; the default and what the GPU kernels compute) and fused multiply-adds (OpenCV's AVX2 + FMA3 dispatch) - as a mask
# with one bit per OpenCV module (oracle.fma_variant.ATAN / .FILTER).  The three f32 checks below try BOTH and record which
# one this OpenCV build computes; `test_report_which_f32_variant_this_opencv_runs` prints the verdict (pytest -rs / -s) and
# fails only if NEITHER form matches.  profiles/r06_fma_risk.json says what changes between them (no histogram bin, no
# oriented point; descriptor entries by <= 5e-7).
F32_VARIANT = {}


def _variants_matching(check):
    found = []
    for mask in (0, 3, 1, 2):
        with oracle.fma_variant(mask):
            if check():
                found.append(mask)
    return found


def test_level_gradients_magnitude_phase(img):
    # processGradients, GaussPyramid.cpp:65-104: Sobel x / y, magnitude, phase(angleInDegrees = true)
    g = oracle.gaussian_blur_u8(img, 0, 1.6)
    gx = cv2.Sobel(g, cv2.CV_32F, 1, 0, ksize=1)
    gy = cv2.Sobel(g, cv2.CV_32F, 0, 1, ksize=1)
    ox, oy, omag, oori = oracle.level_gradients(g)
    assert (gx == ox).all() and (gy == oy).all()
    assert (cv2.magnitude(gx, gy) == omag).all()
    want = cv2.phase(gx, gy, angleInDegrees=True)
    F32_VARIANT["phase"] = [m & 1 for m in _variants_matching(lambda: (oracle.level_gradients(g)[3] == want).all()) if m in (0, 1)]
    assert F32_VARIANT["phase"], "cv::phase matches neither the rounded nor the fused polynomial of the oracle"


def test_float_gaussian_kernel():
    # getGaussianKernel behind GaussianBlur(CV_32F, Size(0,0), sigma): Diff_of_Gauss.cpp:348,618
    for o in range(4):
        for l in range(1, 4):
            sigma = 1.5 * oracle.sigma_at(1.6, o, l)
            n = oracle.gauss_ksize_f32(sigma)
            assert n == (int(round(sigma * 8 + 1)) | 1)
            assert (cv2.getGaussianKernel(n, sigma, cv2.CV_32F).ravel() == oracle.gauss_kernel_f32(n, sigma)).all(), (o, l)


def test_isolated_float_blur_of_a_16x16_window(img):
    # GaussianBlur(magROI, magWeighted, Size(0,0), sigma, 0, BORDER_DEFAULT) on the descriptor's own 16x16
    # Mat, Diff_of_Gauss.cpp:618 (row filter then symmetric column filter, reflect-101 inside the window).
    # The filterKeypoints blur (:348) runs on a ROI of a larger Mat and reads the parent; a numpy view
    # does not carry the parent to cv2, so only the isolated form can be checked from Python.
    _, _, mag, _ = oracle.level_gradients(oracle.gaussian_blur_u8(img, 0, 2.0))
    win = np.ascontiguousarray(mag[40:56, 60:76])
    for sigma in (1.5 * 2.0159, 1.5 * 6.4, 1.5 * 25.6):
        n = oracle.gauss_ksize_f32(sigma)
        k = oracle.gauss_kernel_f32(n, sigma)
        R = n // 2
        ext = win
        while ext.shape[0] < 16 + 2 * R:
            ext = np.pad(ext, min(R - (ext.shape[0] - 16) // 2, ext.shape[0] - 1), mode="reflect")
        off = (ext.shape[0] - 16) // 2 - R
        ext = ext[off:off + 16 + 2 * R, off:off + 16 + 2 * R]
        rowf = (k[0] * ext[:, 0:16]).astype(np.float32)
        for i in range(1, n):
            rowf = (rowf + (k[i] * ext[:, i:i + 16]).astype(np.float32)).astype(np.float32)
        colf = (k[R] * rowf[R:R + 16]).astype(np.float32)
        for i in range(1, R + 1):
            colf = (colf + (k[R + i] * (rowf[R + i:R + i + 16] + rowf[R - i:R - i + 16]).astype(np.float32)).astype(np.float32)).astype(np.float32)
        got = cv2.GaussianBlur(win, (0, 0), sigma, sigmaY=0, borderType=cv2.BORDER_DEFAULT)
        with oracle.fma_variant(False):
            assert (oracle.blur_f32_roi(win, 0, 0, 16, 16, sigma) == colf).all(), sigma  # the oracle's baseline IS this numpy restatement
        ok = [m >> 1 for m in _variants_matching(lambda: (oracle.blur_f32_roi(win, 0, 0, 16, 16, sigma) == got).all()) if m in (0, 2)]
        assert ok, ("cv::GaussianBlur(CV_32F) matches neither the rounded nor the fused filter of the oracle", sigma)
        F32_VARIANT.setdefault("filter_isolated", []).append(ok)


def test_feature_point_localization_matrix_calls():
    # FeaturePointLocalization, Diff_of_Gauss.cpp:233-246, with OpenCV's own gemm / invert
    rng = np.random.default_rng(7)
    cases = [(0, 0, 0, 9), (3, 0, 0, 7), (0, -4, 2, 8), (1, 1, 1, 8), (2, -3, 5, 12), (17, -20, 33, 40)]
    cases += [tuple(int(v) for v in rng.integers(-60, 61, 3)) + (int(rng.integers(0, 256)),) for _ in range(300)]
    for dx, dy, ds, value in cases:
        A = (np.array([[dx], [dy], [ds]], np.float32) / np.float32(255.0)).astype(np.float32)
        A_T = cv2.transpose(A)
        B = cv2.gemm(A, A_T, 1.0, None, 0.0)
        B_inverse = -cv2.invert(B, flags=cv2.DECOMP_LU)[1]
        z_hat = cv2.gemm(B_inverse, A, 1.0, None, 0.0)
        dog_zhat = np.float32(np.float32(value) / np.float32(255.0) + cv2.gemm(A_T, z_hat, 0.5, None, 0.0)[0, 0])
        keep, nv = oracle.feature_point_localization(dx, dy, ds, value)
        assert keep == bool(dog_zhat > np.float32(0.03)), (dx, dy, ds, value)
        if keep:
            x = np.float32(dog_zhat * np.float32(255.0))
            want = int(x) if -2147483904.0 < x < 2147483648.0 else -(2 ** 31)
            assert nv == want, (dx, dy, ds, value)


# ---- the three call sites cv2 cannot reach from Python (VERDICT r4 "What's missing" #2): through the C++ harness
# tools/opencv_pin/pin_harness, built by tools/pin_with_opencv.sh where a C++ OpenCV exists

def _harness():
    import os

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "opencv_pin", "pin_harness")
    if not os.path.exists(exe):
        pytest.skip("tools/opencv_pin/pin_harness is not built (tools/pin_with_opencv.sh builds it against a C++ OpenCV)")
    return exe


def test_roi_blur_reads_the_parent_mat(img, tmp_path):
    # GaussianBlur(windowMag, magWeighted, Size(0,0), sigma, 0, BORDER_DEFAULT) with windowMag = Mat(paddedMag, Rect(x, y, 16, 16)),
    # Diff_of_Gauss.cpp:341-348: no BORDER_ISOLATED, so the filter reads the parent around the window and reflects at the PARENT's
    # edges only - windows in the interior, at a corner and against each edge, kernels narrower and wider than the parent's margin
    import subprocess

    exe = _harness()
    _, _, mag, _ = oracle.level_gradients(oracle.gaussian_blur_u8(img, 0, 2.0))
    parent = np.pad(mag, 8, mode="edge").astype(np.float32)  # GaussPyramid::padOctave(8, octaveGradMag), Diff_of_Gauss.cpp:319 = copyMakeBorder BORDER_REPLICATE, GaussPyramid.cpp:137
    rows, cols = parent.shape
    parent.tofile(tmp_path / "parent.f32")
    cases = []
    for sigma in (1.5 * 2.0159, 1.5 * 6.4, 1.5 * 25.6):
        for x, y in ((40, 50), (0, 0), (cols - 16, rows - 16), (0, rows // 2), (cols // 2, 0), (cols - 16, 3)):
            cases.append((x, y, sigma))
    args = [exe, "roi_blur", str(tmp_path / "parent.f32"), str(rows), str(cols), str(tmp_path / "out.f32")]
    for x, y, s in cases:
        args += [str(x), str(y), repr(s)]
    subprocess.run(args, check=True)
    got = np.fromfile(tmp_path / "out.f32", np.float32).reshape(len(cases), 16, 16)
    for i, (x, y, s) in enumerate(cases):
        ok = [m >> 1 for m in _variants_matching(lambda: (oracle.blur_f32_roi(parent, x, y, 16, 16, s) == got[i]).all()) if m in (0, 2)]
        assert ok, ("the ROI blur matches neither f32 variant of the oracle", x, y, s)
        F32_VARIANT.setdefault("filter_roi", []).append(ok)


def test_report_which_f32_variant_this_opencv_runs():
    # last in the file on purpose.  One verdict per OpenCV module: [0] = the rounded form (the default of the oracle and of the
    # GPU kernels: rows (f)1 / (f)3 / (f)4 are then bit-exact claims), [1] = the fused form (those rows then hold within the
    # tolerance of BASELINE.md section 5), [0, 1] = this input cannot tell them apart
    if not F32_VARIANT:
        pytest.skip("the f32 checks did not run")
    phase = F32_VARIANT.get("phase")
    filt = [set(c) for k in ("filter_isolated", "filter_roi") for c in F32_VARIANT.get(k, [])]
    common = sorted(set.intersection(*filt)) if filt else None
    print(f"\nOPENCV_F32_VARIANT phase={phase} filter={common} (0 = every op rounded, 1 = fused multiply-add)")
    assert common is None or common, "the f32 filter matches a different variant from case to case: " + repr(F32_VARIANT)


def test_determinant_and_trace_double_route(tmp_path):
    # float det = determinant(M); float tr = trace(M)[0]; response = det - k * (tr * tr), Harris_corners.cpp:54-57: both calls
    # compute in double on the CV_32F entries and the assignments narrow to float.  Gradients large enough that the f32 and the
    # f64 routes differ; window 1 makes the structure matrix of the oracle's entry point exactly [ix^2, ix*iy; ix*iy, iy^2].
    import subprocess

    exe = _harness()
    rng = np.random.default_rng(11)
    ix = np.concatenate([rng.integers(-255, 256, 4000).astype(np.float32), (rng.standard_normal(4000) * 3000).astype(np.float32)])
    iy = np.concatenate([rng.integers(-255, 256, 4000).astype(np.float32), (rng.standard_normal(4000) * 3000).astype(np.float32)])
    ix.tofile(tmp_path / "ix.f32")
    iy.tofile(tmp_path / "iy.f32")
    subprocess.run([exe, "det_trace", str(tmp_path / "ix.f32"), str(tmp_path / "iy.f32"), str(len(ix)), "0.04", str(tmp_path / "out.f32")], check=True)
    got = np.fromfile(tmp_path / "out.f32", np.float32).reshape(-1, 3)
    want = oracle.harris_from_grad(ix.reshape(1, -1), iy.reshape(1, -1), 0.04, 1).ravel()  # max(response, 0), :60-62
    assert (np.maximum(got[:, 2], np.float32(0)) == want).all()


def test_mat_at_with_the_column_beyond_the_row(tmp_path):
    # levelImg.at<uchar>(rotatedPoint.x, rotatedPoint.y) on the 20-padded level image, Diff_of_Gauss.cpp:549,775: x is used as the
    # ROW; a release-build Mat::at checks nothing, so the sample is linear element x * (cols + 40) + y of the continuous
    # padded Mat ("SIFT descriptor stage") - the rule behind the oracle's defined / undefined keypoints
    import subprocess

    exe = _harness()
    rows, cols, pad = 37, 61, 20
    img = np.fromfunction(lambda r, c: (r * 131 + c * 7 + (r * c) % 13) % 256, (rows, cols), dtype=np.int64).astype(np.uint8)
    padded = np.pad(img, pad, mode="edge")
    pc = cols + 2 * pad
    pts = [(3, 5), (10, pc + 4), (0, 3 * pc - 1), (rows + 2 * pad - 2, pc + 7), (20, 150)]
    args = [exe, "mat_at", str(rows), str(cols), str(pad), str(tmp_path / "out.u8")]
    for x, y in pts:
        assert x * pc + y < padded.size
        args += [str(x), str(y)]
    subprocess.run(args, check=True)
    got = np.fromfile(tmp_path / "out.u8", np.uint8)
    assert (got == np.array([padded.ravel()[x * pc + y] for x, y in pts], np.uint8)).all()


def test_determinant_and_trace_from_python():
    # the same double route through cv2 itself (no harness needed)
    rng = np.random.default_rng(5)
    for _ in range(500):
        a, b = (rng.standard_normal(2) * 3000).astype(np.float32)
        M = np.array([[a * a, a * b], [a * b, b * b]], np.float32)
        det = np.float32(cv2.determinant(M))
        tr = np.float32(cv2.trace(M)[0])
        resp = np.float32(det - np.float32(np.float32(0.04) * np.float32(tr * tr)))
        want = oracle.harris_from_grad(np.array([[a]], np.float32), np.array([[b]], np.float32), 0.04, 1)[0, 0]
        assert max(resp, np.float32(0)) == want
