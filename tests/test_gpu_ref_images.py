"""The reference's own sample images (tests/golden/ref_images) through the HIP path, bit-exact
against the CPU oracle: the per-image C ABI (what the C++ mirror's Harris / DoG executables call)
and the device-resident batch (vslam_detect_batch_dev) in its three list modes.

chessboard.png is what Harris_corners.cpp:148 loads, home.jpg what Diff_of_Gauss.cpp:730 loads,
building.jpg what tests/GaussPyramid_Test.cpp:78 loads, blox.jpg what tests/rotate_image_test.cpp:25
loads.  Natural content reaches what the synthetic frames do not: Harris responses far beyond the
8-bit view (and beyond 2^31 on the chessboard), smooth DoG stacks with few candidates above the
contrast floor, widths that are not multiples of 8 (868, 1754 -> 877 -> 438).
"""
import numpy as np
import pytest

import oracle
from tests import refimg
from tests.test_gpu_batch import check_frame, run_batch
from visualslam_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    capi.build()
    ctx = capi.Context(0)
    yield ctx, torch
    ctx.close()


@pytest.mark.parametrize("name", refimg.NAMES)
def test_harris_per_image_api(env, name):
    ctx, _ = env
    img = refimg.load(name)
    R = oracle.harris_response(img)
    got = ctx.harris_response(img)
    assert got.tobytes() == R.tobytes()
    u8 = oracle.convert_scale_abs(R)
    assert (ctx.convert_scale_abs(got) == u8).all()
    assert (ctx.nms_strict(u8, 3) == oracle.nms_strict(u8, 3)).all()
    want_n2, want_max = oracle.nms2(R, 5)
    got_n2, got_max = ctx.nms2(got, 5)
    assert got_n2.tobytes() == want_n2.tobytes() and got_max == want_max
    kps = oracle.harris_keypoints(want_n2)
    gk, n = ctx.harris_keypoints(img)
    assert n == len(kps) and gk.tobytes() == kps.tobytes()
    assert (R >= 255.5).sum() > 1000  # the 8-bit view saturates all over a natural image
    if name == "chessboard":  # and wraps to 0 from 2^31 on (x86 cvRound): those survivors are no keypoints
        assert (want_n2 >= np.float32(2.0 ** 31)).any() and (kps["response"] < np.float32(2.0 ** 31)).all()


@pytest.mark.parametrize("name", refimg.NAMES)
def test_dog_per_image_api(env, name):
    # GaussPyramid(img, 4, 1.6) -> initialKeypointDetection -> filterKeypoints, octave by octave,
    # as Diff_of_Gauss.cpp:746,780-787 runs them
    ctx, _ = env
    img = refimg.load(name)
    want, got = oracle.Pyramid(img, 4, 1.6), ctx.pyramid(img, 4, 1.6)
    assert got.sizes == want.sizes == refimg.OCTAVE_SIZES[name]
    for o in range(4):
        assert (got.base(o) == want.base(o)).all(), ("base", o)
        for l in range(6):
            assert (got.gauss(o, l) == want.gauss(o, l)).all(), ("gauss", o, l)
        for l in range(5):
            assert (got.dog(o, l) == want.dog(o, l)).all(), ("dog", o, l)
        wm, wp = want.extrema(o, 3, 8)
        gm, gp, n = got.extrema(o, 3, 8)
        assert (gm == wm).all() and n == len(wp) and gp.tobytes() == wp.tobytes(), ("extrema", o)
        kp = want.keypoints(o, 3)
        gk, nk = got.keypoints(o, 3)
        assert nk == len(kp) and gk.tobytes() == kp.tobytes(), ("keypoints", o)
        w = want.filter_keypoints(o, kp)
        g, nf = got.filter_keypoints(o, gk)
        assert nf == len(w) and g.tobytes() == w.tobytes(), ("oriented", o)
    gx, gy, mag, ori = got.gradients(1, 0)  # tests/GaussPyramid_Test.cpp:114-117 shows octave 1 level 0
    for a, b in zip((gx, gy, mag, ori), oracle.level_gradients(want.gauss(1, 0))):
        assert a.tobytes() == b.tobytes()
    got.close()
    want.close()


@pytest.mark.parametrize("name", refimg.NAMES)
@pytest.mark.parametrize("mode", ["candidates", "localized", "oriented"])
def test_batch_path_on_reference_images(env, name, mode):
    # the image, its transpose-free variants (flipped left-right / upside down) as one batch
    ctx, torch = env
    img = refimg.load(name)
    frames = np.stack([img, img[:, ::-1], img[::-1, :]]).copy()
    kw = dict(localize=int(mode != "candidates"), orient=int(mode == "oriented"))
    p, L, out = run_batch(ctx, torch, frames, with_nms2=(name != "chessboard"), n_octaves=4, **kw)
    for f in range(3):
        check_frame(p, L, out, f, frames[f], 4)
    assert out["harris_counts"].min() > 0 and out["dog_counts"].min() > 0


def test_batch_auto_octaves_building_and_chessboard(env):
    # GaussPyramid(Mat&, sigma) derives 5 and 6 octaves for these two (GaussPyramid.cpp:150-152)
    ctx, torch = env
    for name in ("building", "chessboard"):
        img = refimg.load(name)
        n_oct = capi.auto_num_octaves(*img.shape)
        assert n_oct == refimg.AUTO_OCTAVES[name]
        p, L, out = run_batch(ctx, torch, img[None].copy(), with_nms2=False, n_octaves=n_oct, localize=1)
        check_frame(p, L, out, 0, img, n_oct)


def test_batch_localization_beyond_the_table(env):
    # FeaturePointLocalization with all three differences non-zero and magnitudes >= 16: the sites the
    # kernels' 16^3 table does not cover (natural images have almost none: test_ref_images_cpu.py
    # reports one on the chessboard).  Random 4x4 blocks of 0 / 255 force hundreds of them.
    from tests.test_ref_images_cpu import candidate_differences

    ctx, torch = env
    rng = np.random.default_rng(1)
    frames = np.stack([np.kron((rng.integers(0, 2, (68, 120)) * 255).astype(np.uint8), np.ones((4, 4), np.uint8)) for _ in range(2)])
    want = oracle.Pyramid(frames[0], 4, 1.6)
    d = candidate_differences(want, 0)
    off = (d[:, :3] != 0).all(axis=1) & (np.abs(d[:, :3]).max(axis=1) >= 16)
    want.close()
    assert off.sum() > 100
    for kw in (dict(localize=1), dict(localize=1, orient=1)):
        p, L, out = run_batch(ctx, torch, frames, n_octaves=4, **kw)
        for f in range(2):
            check_frame(p, L, out, f, frames[f], 4)
    got = ctx.pyramid(frames[0], 4, 1.6)
    w2 = oracle.Pyramid(frames[0], 4, 1.6)
    for o in range(4):
        kp = w2.keypoints(o, 3)
        gk, nk = got.keypoints(o, 3)
        assert nk == len(kp) and gk.tobytes() == kp.tobytes()
    got.close()
    w2.close()
