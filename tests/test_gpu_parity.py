"""GPU parity tests: every result of the HIP path, obtained through the C ABI
(include/vslam.h), is compared with the CPU oracle on the same seeded inputs.

Bar: bit-exact for every integer output (blurred / Gaussian / DoG
images, masks, bitmasks, keypoint lists) AND bit-exact for the f32 Harris response (stated
tolerance: 0 ulp -- the kernels reproduce the reference's rounding sequence exactly).
"""
import os

import numpy as np
import pytest

import oracle
from visualslam_amd import capi, synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SHAPES = [(48, 64), (33, 47), (1, 1), (1, 9), (7, 1), (2, 2), (16, 64), (17, 65), (70, 130), (3, 300), (1, 4), (2, 8), (5, 12), (9, 244), (300, 4)]


@pytest.fixture(scope="module")
def ctx():
    capi.build()
    c = capi.Context(0)
    yield c
    c.close()


def frame(shape, kind="noise", sid=0):
    return synth.frame_np(shape[0], shape[1], 0, sid, kind)


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


# ---------------------------------------------------------------- primitives


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("ksize,sigma", [(3, 0.0), (0, 1.6), (0, 5.079683366298239), (7, 0.0), (5, 1.1), (0, 12.8)])
def test_gaussian_blur(ctx, shape, ksize, sigma):
    img = frame(shape)
    assert same(ctx.gaussian_blur_u8(img, ksize, sigma), oracle.gaussian_blur_u8(img, ksize, sigma))


def test_gaussian_blur_widest_reference_kernel(ctx):
    # octave-3 level-5 of a 1080p pyramid: 245 taps on a 270x480 image
    img = frame((270, 480), "checker")
    s = oracle.sigma_at(1.6, 3, 5)
    assert oracle.gauss_ksize_u8(s) == 245
    assert same(ctx.gaussian_blur_u8(img, 0, s), oracle.gaussian_blur_u8(img, 0, s))
    tiny = frame((20, 30))  # radius 122 >> image: repeated reflection
    assert same(ctx.gaussian_blur_u8(tiny, 0, s), oracle.gaussian_blur_u8(tiny, 0, s))


def test_blur_strided_views_and_errors(ctx):
    big = frame((40, 80))
    view = big[:, 8:50]  # non-contiguous rows go through capi as a copy; step handling is in C
    assert same(ctx.gaussian_blur_u8(view, 0, 2.0), oracle.gaussian_blur_u8(view, 0, 2.0))
    with pytest.raises(capi.VslamError):
        ctx.gaussian_blur_u8(big, 4, 1.0)  # even kernel
    with pytest.raises(capi.VslamError):
        ctx.gaussian_blur_u8(big, 0, 0.0)  # no size, no sigma
    with pytest.raises(capi.VslamError):
        ctx.sobel_k1(big, 1, 1)


@pytest.mark.parametrize("shape", SHAPES)
def test_sobel_resize_convert(ctx, shape):
    img = frame(shape)
    assert same(ctx.sobel_k1(img, 1, 0), oracle.sobel_k1(img, 1, 0))
    assert same(ctx.sobel_k1(img, 0, 1), oracle.sobel_k1(img, 0, 1))
    assert same(ctx.resize_linear2x(img), oracle.resize_linear2x(img))
    if shape[0] > 1 and shape[1] > 1:
        assert same(ctx.resize_nearest_half(img), oracle.resize_nearest_half(img))


def test_convert_scale_abs(ctx):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((37, 53)) * 300).astype(np.float32)
    x[0, :14] = [0.5, 1.5, 2.5, 253.5, 254.5, 255.5, 1e12, -1e12, 2147483520.0, 2147483648.0, -2147483648.0, np.nan, np.inf, 1e9]
    assert same(ctx.convert_scale_abs(x), oracle.convert_scale_abs(x))


# --------------------------------------------------------------------- Harris


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("kind", ["noise", "checker"])
def test_harris_response_fused_bit_exact(ctx, shape, kind):
    img = frame(shape, kind)
    assert same(ctx.harris_response(img), oracle.harris_response(img))


@pytest.mark.parametrize("window", [1, 3, 5])
def test_harris_from_arbitrary_float_gradients_bit_exact(ctx, window):
    rng = np.random.default_rng(11)
    ix = (rng.standard_normal((41, 67)) * 37.123).astype(np.float32)
    iy = (rng.standard_normal((41, 67)) * 91.7).astype(np.float32)
    got = ctx.harris_from_grad(ix, iy, 0.04, window)
    assert same(got, oracle.harris_from_grad(ix, iy, 0.04, window))
    assert (got >= 0).all()


def test_harris_stagewise_equals_fused(ctx):
    img = frame((75, 101), "checker")
    b = ctx.gaussian_blur_u8(img, 3, 0.0)
    r = ctx.harris_from_grad(ctx.sobel_k1(b, 1, 0), ctx.sobel_k1(b, 0, 1))
    assert same(r, ctx.harris_response(img))


@pytest.mark.parametrize("shape", [(48, 64), (33, 47), (1, 1), (4, 4), (5, 9), (70, 130)])
def test_nms_variants(ctx, shape):
    img = frame(shape, "checker")
    R = oracle.harris_response(img)
    u8 = oracle.convert_scale_abs(R)
    for w in (1, 3, 5):
        assert same(ctx.nms_strict(u8, w), oracle.nms_strict(u8, w))
        assert same(ctx.nms_strict(R, w), oracle.nms_strict(R, w))
    for w in (1, 3, 5, 7):
        got, tm = ctx.nms2(R, w)
        want, wtm = oracle.nms2(R, w)
        assert same(got, want) and tm == wtm
    with pytest.raises(capi.VslamError):
        ctx.nms_strict(u8, 4)


def test_nms_plateaus_and_ties(ctx):
    R = np.zeros((12, 14), np.float32)
    R[4:7, 4:7] = 300.0  # plateau: '>=' keeps every member whose half-open window max equals it
    R[9, 9] = 254.0
    got, _ = ctx.nms2(R, 5)
    want, _ = oracle.nms2(R, 5)
    assert same(got, want)
    u8 = oracle.convert_scale_abs(R)
    assert same(ctx.nms_strict(u8, 3), oracle.nms_strict(u8, 3))


@pytest.mark.parametrize("shape,kind", [((48, 64), "checker"), ((96, 160), "checker"), ((70, 130), "noise"), ((4, 70), "noise")])
def test_harris_keypoint_list_identical(ctx, shape, kind):
    img = frame(shape, kind)
    want = oracle.harris_keypoints(oracle.nms2(oracle.harris_response(img), 5)[0])
    got, n = ctx.harris_keypoints(img)
    assert n == len(want) and same(got, want)
    if len(want) > 3:  # capacity smaller than the list: prefix + full count
        got2, n2 = ctx.harris_keypoints(img, cap=3)
        assert n2 == len(want) and same(got2, want[:3])


# ---------------------------------------------------------------- DoG pyramid


def check_pyramid(ctx, img, n_oct, sigma0=1.6):
    want = oracle.Pyramid(img, n_oct, sigma0)
    got = ctx.pyramid(img, n_oct, sigma0)
    try:
        assert got.n_octaves == want.n_octaves and got.sizes == want.sizes
        assert got.sigmas == want.sigmas and got.ksizes == want.ksizes
        for o in range(n_oct):
            assert same(got.base(o), want.base(o)), ("base", o)
            for l in range(6):
                assert same(got.gauss(o, l), want.gauss(o, l)), ("gauss", o, l)
            for l in range(5):
                assert same(got.dog(o, l), want.dog(o, l)), ("dog", o, l)
            for mc in (8, 0, 3):
                wm, wp = want.extrema(o, 3, mc)
                gm, gp, n = got.extrema(o, 3, mc)
                assert same(gm, wm), ("mask", o)
                assert n == len(wp) and same(gp, wp), ("points", o, mc)
            wk = want.keypoints(o, 3)
            gk, nk = got.keypoints(o, 3)
            assert nk == len(wk) and same(gk, wk), ("localized keypoints", o)
    finally:
        got.close()
        want.close()


@pytest.mark.parametrize("shape,n_oct", [((48, 64), 3), ((33, 47), 3), ((40, 56), 4), ((9, 13), 2), ((2, 3), 1), ((135, 240), 4), ((4, 8), 3), ((16, 32), 4), ((3, 64), 2), ((67, 4), 2)])
@pytest.mark.parametrize("kind", ["checker", "noise"])
def test_pyramid_and_extrema_bit_exact(ctx, shape, n_oct, kind):
    check_pyramid(ctx, frame(shape, kind, 4), n_oct)


def test_pyramid_other_sigma_and_window(ctx):
    img = frame((50, 70), "checker", 2)
    check_pyramid(ctx, img, 2, 1.2)
    want, got = oracle.Pyramid(img, 2, 1.6), ctx.pyramid(img, 2, 1.6)
    wm, wp = want.extrema(0, 5, 8)  # windowSize 5: pad 2, 4x4x3 window, stride 5
    gm, gp, n = got.extrema(0, 5, 8)
    assert same(gm, wm) and n == len(wp) and same(gp, wp)
    for o in range(2):
        wk = want.keypoints(o, 5)
        gk, nk = got.keypoints(o, 5)
        assert nk == len(wk) and same(gk, wk)
    with pytest.raises(capi.VslamError):
        got.extrema(0, 4, 8)
    with pytest.raises(capi.VslamError):
        got.extrema(7, 3, 8)
    with pytest.raises(capi.VslamError):
        got.gauss(0, 6)


def test_pyramid_auto_octaves_and_constant_image(ctx):
    img = synth.frame_np(64, 80, kind="constant")
    p = ctx.pyramid(img, 0, 1.6)  # second constructor: floor(log2(64)) - 4 = 2
    assert p.n_octaves == 2 == oracle.auto_num_octaves(64, 80)
    for o in range(2):
        for l in range(5):
            assert not p.dog(o, l).any()
        m, pts, n = p.extrema(o, 3, 8)
        assert m.all() and n == 0  # every lattice site is a (value 0) candidate; none passes contrast
        m0, pts0, n0 = p.extrema(o, 3, 0)
        assert n0 == m0.size == len(pts0)


def test_impulse_dog_equals_kernel_difference(ctx):
    # single white pixel: G_l = (255*t_l (x) t_l + 32768) >> 16 on the 2x-upsampled image;
    # compare against the oracle and check the DoG is the saturating difference
    img = synth.frame_np(33, 33, kind="impulse")
    p = ctx.pyramid(img, 1, 1.6)
    g = [p.gauss(0, l).astype(int) for l in range(6)]
    for l in range(5):
        assert (p.dog(0, l) == np.maximum(g[l + 1] - g[l], 0)).all()
    check_pyramid(ctx, img, 1)


@pytest.mark.parametrize("name", ["checker_48x64", "noise_40x56", "checker_33x47"])
def test_golden_fixtures(ctx, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    img = g["img"]
    R = ctx.harris_response(img)
    assert same(R, g["response"])
    assert same(ctx.nms_strict(ctx.convert_scale_abs(R), 3), g["nms_mask"])
    n2, tm = ctx.nms2(R, 5)
    assert same(n2, g["nms2"]) and np.float32(tm) == g["nms2_true_max"]
    kps, n = ctx.harris_keypoints(img)
    assert n == len(g["harris_kps"]) and same(kps, g["harris_kps"])
    p = ctx.pyramid(img, int(g["n_octaves"]), 1.6)
    for o in range(p.n_octaves):
        assert same(p.base(o), g[f"base_{o}"])
        assert same(np.stack([p.gauss(o, l) for l in range(6)]), g[f"gauss_{o}"])
        assert same(np.stack([p.dog(o, l) for l in range(5)]), g[f"dog_{o}"])
        m, pts, cnt = p.extrema(o, 3, 8)
        assert same(m, g[f"ext_mask_{o}"]) and cnt == len(g[f"ext_pts_{o}"]) and same(pts, g[f"ext_pts_{o}"])
        kp, nk = p.keypoints(o, 3)
        assert nk == len(g[f"kp_pts_{o}"]) and same(kp, g[f"kp_pts_{o}"])
        fk, nf = p.filter_keypoints(o, kp)
        assert nf == len(g[f"oriented_pts_{o}"]) and same(fk, g[f"oriented_pts_{o}"])
        desc, ok = p.sift_descriptors(o, fk)
        assert (ok == g[f"sift_defined_{o}"]).all() and same(np.nan_to_num(desc, nan=-1.0), g[f"sift_desc_{o}"])


def test_feature_point_localization_bit_exact(ctx):
    # the singular-matrix contrast test must round like the oracle for
    # every combination, including the ones whose outcome is rounding noise
    r = np.arange(-12, 13)
    grid = np.stack(np.meshgrid(r, r, r, [0, 5, 7, 8, 9, 40, 255], indexing="ij"), -1).reshape(-1, 4)
    # both sides of the edge of the 16^3 table of the quadratic term (kernels_localize.hip.h)
    e = np.array([-17, -16, -15, -14, -1, 1, 14, 15, 16, 17])
    edge = np.stack(np.meshgrid(e, e, e, [0, 8, 200], indexing="ij"), -1).reshape(-1, 4)
    rng = np.random.default_rng(7)
    rnd = np.concatenate([rng.integers(-255, 256, (200000, 3)), rng.integers(0, 256, (200000, 1))], 1)
    d = np.concatenate([grid, edge, rnd]).astype(np.int32)
    keep, val = ctx.localize_points(d)
    wk = np.zeros(len(d), bool)
    wv = d[:, 3].copy()
    for i, (a, b, c, v) in enumerate(d.tolist()):
        k, nv = oracle.feature_point_localization(a, b, c, v)
        wk[i] = k
        if k:
            wv[i] = nv
    assert (keep == wk).all(), int((keep != wk).sum())
    assert (val == wv).all(), int((val != wv).sum())
    three = (d[:, :3] != 0).all(1)
    assert 0 < keep[three].sum() < three.sum()  # the noisy branch is exercised both ways
    k0, v0 = ctx.localize_points(np.zeros((0, 4), np.int32))
    assert len(k0) == 0 and len(v0) == 0


@pytest.mark.parametrize("shape,n_oct,kind", [((96, 160), 3, "noise"), ((75, 131), 2, "noise"), ((48, 64), 3, "checker"), ((135, 240), 4, "noise"), ((20, 24), 2, "noise")])
def test_filter_keypoints_bit_exact(ctx, shape, n_oct, kind):
    # edge rejection + blurred-magnitude orientation histogram; the list
    # (positions, angles, order) must equal the oracle's.  Small images make the 1.5*sigma blur
    # kernels wider than the padded image (multiple reflections).
    img = frame(shape, kind, 6)
    want, got = oracle.Pyramid(img, n_oct, 1.6), ctx.pyramid(img, n_oct, 1.6)
    total = 0
    for o in range(n_oct):
        kp = want.keypoints(o, 3)
        w = want.filter_keypoints(o, kp)
        g, n = got.filter_keypoints(o, kp)
        assert n == len(w) and same(g, w), (o, n, len(w))
        g2, n2 = got.filter_keypoints(o, kp, cap=3)
        assert n2 == len(w) and same(g2, w[:3])
        total += n
        # synthetic keypoints at the corners / edges of the padded coordinate range, all six levels
        r, c = want.sizes[o]
        extra = np.zeros(12, capi.POINT_DTYPE)
        for i, (y, x) in enumerate([(0, 0), (r, c), (0, c), (r, 0), (1, 1), (r - 1, c - 1), (r // 2, 0), (0, c // 2), (r // 2, c // 2), (1, c), (r, 1), (2, 3)]):
            extra[i] = (y, x, 50, 1, o, i % 6)
        w = want.filter_keypoints(o, extra)
        g, n = got.filter_keypoints(o, extra)
        assert n == len(w) and same(g, w), ("extra", o)
    assert total > 0 or kind == "checker"
    e, n = got.filter_keypoints(0, np.zeros(0, capi.POINT_DTYPE))
    assert n == 0 and len(e) == 0
    bad = np.zeros(1, capi.POINT_DTYPE)
    bad[0] = (1, 1, 0, 1, 0, 6)
    with pytest.raises(capi.VslamError):
        got.filter_keypoints(0, bad)
    with pytest.raises(ValueError):
        want.filter_keypoints(0, bad)
    got.close()


def test_edge_response_windows(ctx):
    rng = np.random.default_rng(5)
    gx = rng.integers(-255, 256, (5000, 4)).astype(np.float32)
    gy = rng.integers(-255, 256, (5000, 4)).astype(np.float32)
    gx[:50] = 0
    gy[50:100] = gx[50:100]  # singular: det == 0 -> inf / nan
    got = ctx.edge_response_windows(gx, gy)
    want = np.array([oracle.compute_edge_response(a.reshape(2, 2), b.reshape(2, 2), 1, 1, 1) for a, b in zip(gx, gy)], np.float32)
    assert got.tobytes() == want.tobytes()
    assert not np.isfinite(got[:100]).any()  # tr^2 / 0
    # StructureMatrix (Harris_corners.cpp:10-29): the same sums over a 3x3 window, entries of M
    wx = rng.integers(-255, 256, (300, 9)).astype(np.float32)
    wy = rng.integers(-255, 256, (300, 9)).astype(np.float32)
    m = ctx.structure_matrix_windows(wx, wy)
    want_m = np.stack([(wx * wx).sum(1), (wx * wy).sum(1), (wy * wy).sum(1)], 1)  # integers < 2^24: exact in f32
    assert m.tobytes() == want_m.astype(np.float32).tobytes()


@pytest.mark.parametrize("shape,n_oct", [((40, 56), 2), ((33, 47), 2), ((1, 5), 1)])
def test_process_gradients(ctx, shape, n_oct):
    # Sobel x/y exact, magnitude and fastAtan2 orientation bit-exact
    # against the oracle's restatement (stated tolerance if ever relaxed: 1e-4 deg / 1 ulp)
    img = frame(shape, "checker", 5)
    want = oracle.Pyramid(img, n_oct, 1.6)
    got = ctx.pyramid(img, n_oct, 1.6)
    for o in range(n_oct):
        for l in (0, 3, 5):
            w = oracle.level_gradients(want.gauss(o, l))
            g4 = got.gradients(o, l)
            for a, b, name in zip(g4, w, ("gx", "gy", "mag", "orient")):
                assert same(a, b), (name, o, l, float(np.abs(a - b).max()), int((a != b).sum()))
            assert g4[3].min() >= 0.0 and g4[3].max() < 360.0
    with pytest.raises(capi.VslamError):
        got.gradients(0, 6)
    got.close()


def test_fast_correctly_rounded_sqrt_is_exhaustively_exact(tmp_path):
    # round 3: the magnitudes (cv::magnitude of integer Sobel differences) use a 9-instruction correctly
    # rounded f32 square root instead of the f64 one; every argument that can occur (0 .. 2*255^2) is compared
    import json
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sqrt_check")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-o", exe, os.path.join(root, "tools", "sqrt_check.hip")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and out["mismatches"] == 0 and out["checked"] == 2 * 255 * 255 + 1, out
