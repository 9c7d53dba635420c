"""SURVEY section 8f row 4 on the GPU: vslam_sift_descriptors (rotateImageSection + SIFT,
Diff_of_Gauss.cpp:528-559,561-693) against the CPU oracle, on the reference's images and on synthetic
frames, including keypoints whose rotated window leaves the padded level (undefined in the reference)."""
import numpy as np
import pytest

import oracle
from tests import refimg
from visualslam_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    capi.build()
    c = capi.Context(0)
    yield c
    c.close()


def both(ctx, img, n_oct=4):
    want, got = oracle.Pyramid(img, n_oct, 1.6), ctx.pyramid(img, n_oct, 1.6)
    stats = []
    for o in range(n_oct):
        oriented = want.filter_keypoints(o, want.keypoints(o, 3))
        wd, wok = want.sift_descriptors(o, oriented)
        gd, gok = got.sift_descriptors(o, oriented)
        assert (gok == wok).all(), o
        assert np.array_equal(gd, wd, equal_nan=True), o  # bit-exact apart from the NaN payload of 0 / 0
        stats.append((len(oriented), int(wok.sum()), int(np.isnan(wd).any(axis=1).sum())))
    got.close()
    want.close()
    return stats


@pytest.mark.parametrize("name", refimg.NAMES)
def test_sift_descriptors_on_reference_images(ctx, name):
    stats = both(ctx, refimg.load(name))
    assert sum(s[0] for s in stats) > 0
    if name == "blox":  # square: every window defined
        assert all(s[0] == s[1] for s in stats)
    if name in ("home", "building", "chessboard"):  # landscape: x used as the row runs off the padded level
        assert any(s[1] < s[0] for s in stats)


@pytest.mark.parametrize("shape,kind", [((96, 96), "noise"), ((200, 120), "noise"), ((120, 200), "checker"), ((64, 64), "checker")])
def test_sift_descriptors_on_synthetic_frames(ctx, shape, kind):
    stats = both(ctx, synth.frame_np(*shape, kind=kind), 3)
    assert sum(s[0] for s in stats) > 0


def test_sift_every_angle_and_the_level_edges(ctx):
    # hand-made oriented keypoints: all 36 histogram angles, positions at the corners of the padded
    # coordinate range, every level the pipeline can produce
    img = synth.frame_np(150, 150, kind="noise")
    want, got = oracle.Pyramid(img, 2, 1.6), ctx.pyramid(img, 2, 1.6)
    for o in range(2):
        r, c = want.sizes[o]
        kps = []
        for a in range(0, 360, 10):
            for (y, x) in ((1, 1), (1, c), (r, 1), (r, c), (r // 2 + 1, c // 2 + 1), (7, c - 5)):
                kps.append((y, x, a, 0, o, 1 + (a // 10) % 3))
        kps = np.array(kps, dtype=oracle.POINT_DTYPE)
        wd, wok = want.sift_descriptors(o, kps)
        gd, gok = got.sift_descriptors(o, kps)
        assert (gok == wok).all() and np.array_equal(gd, wd, equal_nan=True)
        assert wok.all()  # square level: x + 12 <= cols + 32 < rows + 40, every window stays inside the padded Mat
    got.close()
    want.close()


def test_sift_undefined_without_flags_is_an_error(ctx):
    img = refimg.load("home")
    got = ctx.pyramid(img, 4, 1.6)
    want = oracle.Pyramid(img, 4, 1.6)
    oriented = want.filter_keypoints(0, want.keypoints(0, 3))
    _, ok = want.sift_descriptors(0, oriented)
    assert not ok.all()
    import ctypes as C

    desc = np.zeros((len(oriented), 128), np.float32)
    rc = capi.lib().vslam_sift_descriptors(ctx._h, got._h, 0, oriented.ctypes.data, len(oriented), desc.ctypes.data, None)
    assert rc == -5  # VSLAM_ERR_RANGE
    bad = oriented[:1].copy()
    bad["level"] = 9
    with pytest.raises(capi.VslamError):
        got.sift_descriptors(0, bad)
    got.close()
    want.close()
