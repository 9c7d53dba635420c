"""The reference's own sample images (tests/golden/ref_images, see tests/refimg.py) through the CPU
oracle: the known answers SURVEY.md section 4 / Appendix C derive for them, the independent
integer restatements of tests/npref.py on natural-image content, and the paths only real content
reaches (Harris responses beyond the 8-bit view and beyond 2^31, localization differences >= 16).

These inputs are reference-held data; the expected outputs still come from the oracle and its
numpy cross-checks (the reference stores no expected values), so parity stays "unpinned" against
a real OpenCV build -- tests/test_opencv_crosscheck.py closes that the day cv2 is importable.
"""
import numpy as np
import pytest

import oracle
from tests import npref, refimg


@pytest.fixture(scope="module")
def images():
    return {n: refimg.load(n) for n in refimg.NAMES}


@pytest.fixture(scope="module")
def pyramids(images):
    pyr = {n: oracle.Pyramid(images[n], 4, 1.6) for n in refimg.NAMES}
    yield pyr
    for p in pyr.values():
        p.close()


def test_fixtures_are_the_reference_images(images):
    # SHA-256 prefixes of the four source files; sizes from section 2.1 row 6
    want = {"blox": ("11115aa2", (256, 256)), "home": ("23b8cf46", (384, 512)),
            "building": ("742a1baa", (600, 868)), "chessboard": ("7f5ce40e", (1240, 1754))}
    for n, (sha, shape) in want.items():
        assert refimg.MANIFEST[n]["source_sha256"].startswith(sha)
        assert images[n].shape == shape and images[n].dtype == np.uint8
        assert images[n].std() > 10  # real content, not a blank


def test_pyramid_known_answers_on_reference_images(images, pyramids):
    # tests/GaussPyramid_Test.cpp:88-110 prints these for building.jpg; Appendix C lists all four
    for n in refimg.NAMES:
        p = pyramids[n]
        assert p.n_octaves == 4 and p.sizes == refimg.OCTAVE_SIZES[n]
        assert oracle.auto_num_octaves(*images[n].shape) == refimg.AUTO_OCTAVES[n]
        assert p.sigmas[0][0] == 1.6 and abs(p.sigmas[0][3] - 3.2) < 1e-15
        assert abs(p.sigmas[0][5] - 5.079683366298239) < 1e-14
        assert p.ksizes[0] == [11, 13, 17, 21, 25, 31] and p.ksizes[3] == [79, 99, 123, 155, 195, 245]
    # chessboard: 877 -> 438 is the round-half-even case of resize(0.5)
    assert pyramids["chessboard"].sizes[2:] == [(620, 877), (310, 438)]


@pytest.mark.parametrize("name", ["blox", "home"])
def test_pyramid_matches_integer_restatement(images, pyramids, name):
    # every Gaussian / DoG image of the small reference images against tests/npref.py:
    # 2x bilinear base, exact 2-D integer sums with the 8.8 taps, saturating DoG, decimation
    p, img = pyramids[name], images[name]
    base = npref.resize2x(img)
    for o in range(4):
        assert (p.base(o) == base).all(), ("base", o)
        g = []
        for l in range(6):
            taps = oracle.gauss_taps_q8(p.ksizes[o][l], p.sigmas[o][l])
            g.append(npref.blur_q8(base, taps))
            assert (p.gauss(o, l) == g[l]).all(), ("gauss", o, l)
        for l in range(5):
            want = np.clip(g[l + 1].astype(np.int16) - g[l], 0, 255).astype(np.uint8)
            assert (p.dog(o, l) == want).all(), ("dog", o, l)
        base = g[3][::2, ::2][: (g[3].shape[0] + 1) // 2, : (g[3].shape[1] + 1) // 2]
        if o < 3:
            assert base.shape == p.sizes[o + 1]
        dogs = [p.dog(o, l) for l in range(5)]
        mask, _, _ = npref.extrema_mask(dogs, 3)
        assert (p.extrema(o, 3, 8)[0] == mask).all(), ("mask", o)


@pytest.mark.parametrize("name", refimg.NAMES)
def test_harris_matches_integer_restatement(images, name):
    img = images[name]
    R = oracle.harris_response(img)
    assert R.tobytes() == npref.harris_int(img).tobytes()
    # natural content leaves the 8-bit range of the u8 view by orders of magnitude
    assert (R >= 255.5).sum() > 1000 and R.max() > 1e8


def test_harris_literal_loops_on_the_square_image(images):
    # Harris_corners.cpp:40,47-50 allocate zeros(width, height) and use (i in cols, j in rows) as
    # (row, col): defined behaviour only for square inputs (SURVEY B-1), where it must equal the
    # intended operator.  blox.jpg (256x256) is the reference's one square image.  The loops below
    # keep the reference's index roles literally (i runs over cols and is used as the ROW).
    img = images["blox"]
    blurred = oracle.gaussian_blur_u8(img, 3, 0.0)
    ix, iy = oracle.sobel_k1(blurred, 1, 0), oracle.sobel_k1(blurred, 0, 1)
    width, height, pad = ix.shape[1], ix.shape[0], 1
    pix, piy = np.pad(ix, pad, mode="edge"), np.pad(iy, pad, mode="edge")
    image = np.zeros((width, height), np.float32)
    k = np.float32(0.04)
    js = np.arange(pad, pix.shape[0] - pad)
    for i in range(pad, pix.shape[1] - pad):
        ix2 = np.zeros(len(js), np.float32)
        iy2 = np.zeros(len(js), np.float32)
        ixy = np.zeros(len(js), np.float32)
        for u in range(i - pad, i + pad + 1):
            for dv in range(-pad, pad + 1):
                a, b = pix[u, js + dv], piy[u, js + dv]
                ix2 += a * a
                iy2 += b * b
                ixy += a * b
        det = (ix2.astype(np.float64) * iy2.astype(np.float64) - ixy.astype(np.float64) * ixy.astype(np.float64)).astype(np.float32)
        tr = (ix2.astype(np.float64) + iy2.astype(np.float64)).astype(np.float32)
        resp = det - k * (tr * tr)
        image[i - pad, js - pad] = np.where(resp > 0, resp, np.float32(0))
    assert image.tobytes() == oracle.harris_response(img).tobytes()


def test_convert_scale_abs_beyond_int32_on_the_chessboard(images):
    # chessboard.png is the image the Harris executable loads (Harris_corners.cpp:148); its
    # strongest corners exceed 2^31, where x86 cvRound (cvtps2dq / cvtss2si) returns INT_MIN and
    # saturate_cast<uchar> maps it to 0 -- the reference's literal behaviour, oracle default
    R = oracle.harris_response(images["chessboard"])
    big = R >= np.float32(2.0 ** 31)
    assert big.sum() > 500
    u8 = oracle.convert_scale_abs(R)
    assert not u8[big].any()
    mid = (R >= 255.5) & ~big
    assert mid.sum() > 1000 and (u8[mid] == 255).all()
    low = R < 255.5
    assert (u8[low] == np.rint(R[low]).astype(np.uint8)).all()
    # the keypoint criterion (:139, abs_NMS > 253) therefore drops NMS2 survivors >= 2^31
    n2, _ = oracle.nms2(R, 5)
    kps = oracle.harris_keypoints(n2)
    assert len(kps) > 0 and (kps["response"] < np.float32(2.0 ** 31)).all() and (kps["response"] >= 253.5).all()
    assert ((n2 >= np.float32(2.0 ** 31)).sum()) > 0


def candidate_differences(p, o, window=3):
    """(d_x, d_y, d_scale, value) of every extrema candidate of octave o (Diff_of_Gauss.cpp:226-228)."""
    pad = (window - 1) // 2
    dogs = [np.pad(p.dog(o, l).astype(np.int32), pad, mode="edge") for l in range(5)]
    mask, pts = p.extrema(o, window, 0)
    if len(pts) == 0:
        return np.zeros((0, 4), np.int32)
    i, j, lv = pts["row"], pts["col"], pts["level"]
    d = np.stack([np.stack(dogs)[lv, i, j] for _ in range(1)])[0]
    D = np.stack(dogs)
    # padded coordinates index the padded images directly; the +1 neighbours clamp like padOctave
    ip, jp = np.minimum(i + 1, D.shape[1] - 1), np.minimum(j + 1, D.shape[2] - 1)
    dx = D[lv, i, j - 1] - D[lv, i, jp]
    dy = D[lv, i - 1, j] - D[lv, ip, j]
    ds = D[lv - 1, i, j] - D[lv + 1, i, j]
    return np.stack([dx, dy, ds, d], axis=1).astype(np.int32)


def test_localization_paths_real_images_reach(pyramids, capsys):
    # FeaturePointLocalization (Diff_of_Gauss.cpp:223-251): with one zero difference the 3x3
    # "inverse" is exactly zero (closed form); with three non-zero differences it is rounding
    # noise ("slow path" of the kernels), and differences >= 16 fall off the kernels' 16^3 table.
    # Natural images must exercise all three; the shares are reported per image.
    total_off = 0
    report = []
    for n in refimg.NAMES:
        p = pyramids[n]
        nc = slow = off = kept = 0
        for o in range(4):
            d = candidate_differences(p, o)
            nz = (d[:, :3] != 0).all(axis=1)
            big = nz & (np.abs(d[:, :3]).max(axis=1) >= 16)
            nc, slow, off = nc + len(d), slow + int(nz.sum()), off + int(big.sum())
            kp = p.keypoints(o, 3)
            kept += len(kp)
            # the oracle's per-candidate function reproduces its own list
            keep = [oracle.feature_point_localization(*map(int, r)) for r in d[nz][:2000]]
            assert len(keep) == min(int(nz.sum()), 2000)
        report.append((n, nc, slow, off, kept))
        total_off += off
        assert slow > 0, n
    with capsys.disabled():
        for n, nc, slow, off, kept in report:
            print(f"\n  {n}: {nc} candidates, {slow} ({100.0 * slow / nc:.2f} %) with three non-zero differences, "
                  f"{off} of them beyond the 16^3 table, {kept} kept", end="")
    assert total_off > 0


def test_reference_images_full_pipeline_counts(images, pyramids):
    # regression pins (oracle outputs on reference-held inputs; see the module docstring)
    got = {}
    for n in refimg.NAMES:
        R = oracle.harris_response(images[n])
        kps = oracle.harris_keypoints(oracle.nms2(R, 5)[0])
        p = pyramids[n]
        cand = sum(len(p.extrema(o, 3, 8)[1]) for o in range(4))
        kp = [p.keypoints(o, 3) for o in range(4)]
        ori = sum(len(p.filter_keypoints(o, kp[o])) for o in range(4))
        got[n] = (len(kps), cand, sum(map(len, kp)), ori)
    assert all(v[0] > 0 and v[1] > 0 and v[2] > 0 and v[3] > 0 for v in got.values()), got
