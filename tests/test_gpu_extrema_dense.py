"""GPU: vslam_dog_extrema_dense (extension, the dense 3x3x3 scale-space test) against the oracle's loop,
bit for bit: candidate bitmask and the ordered list."""
import numpy as np
import pytest

import oracle
from visualslam_amd import capi, synth

from tests import refimg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    capi.build()
    c = capi.Context(0)
    yield c
    c.close()


def check(ctx, img, n_oct, mcs=(8, 0, 40)):
    want, got = oracle.Pyramid(img, n_oct), ctx.pyramid(img, n_oct)
    try:
        for o in range(n_oct):
            for mc in mcs:
                wm, wp = want.extrema_dense(o, mc)
                gm, gp, n = got.extrema_dense(o, mc)
                assert gm.shape == wm.shape and gm.tobytes() == wm.tobytes(), ("mask", o, mc)
                assert n == len(wp) and gp.tobytes() == wp.tobytes(), ("points", o, mc)
    finally:
        got.close()
        want.close()


# widths around the quad / 64-pixel word / 256-quad workgroup seams, heights around the 32-row segments
@pytest.mark.parametrize("shape,n_oct", [((48, 64), 3), ((33, 47), 2), ((1, 1), 1), ((2, 3), 1), ((5, 1), 1), ((70, 130), 2), ((9, 515), 1),
                                         ((31, 32), 2), ((65, 33), 2), ((3, 1030), 1), ((100, 255), 2), ((135, 240), 4)])
@pytest.mark.parametrize("kind", ["noise", "checker"])
def test_dense_extrema_bit_exact(ctx, shape, n_oct, kind):
    check(ctx, synth.frame_np(shape[0], shape[1], 0, 5, kind), n_oct)


def test_dense_extrema_constant_image_every_pixel_is_a_candidate(ctx):
    got = ctx.pyramid(np.full((40, 70), 91, np.uint8), 2)
    try:
        for o in range(2):
            m, pts, n = got.extrema_dense(o, 0)
            assert m.all() and n == m.size and (pts["value"] == 0).all()
            m, pts, n = got.extrema_dense(o, 1)
            assert m.all() and n == 0
    finally:
        got.close()


@pytest.mark.parametrize("name", ["blox", "home"])
def test_dense_extrema_reference_images(ctx, name):
    check(ctx, refimg.load(name), 3, mcs=(8,))


def test_dense_extrema_list_cap_and_errors(ctx):
    img = synth.frame_np(60, 80, 0, 1, "noise")
    want, got = oracle.Pyramid(img, 1), ctx.pyramid(img, 1)
    try:
        _, wp = want.extrema_dense(0, 0)
        _, gp, n = got.extrema_dense(0, 0, cap=100)
        assert n == len(wp) > 100 and len(gp) == 100 and gp.tobytes() == wp[:100].tobytes()
        with pytest.raises(capi.VslamError):
            got.extrema_dense(1, 8)
    finally:
        got.close()
        want.close()


def test_dense_extrema_full_1080p(ctx):
    img = synth.frame_np(1080, 1920, 0, 2, "checker")
    check(ctx, img, 2, mcs=(8,))


@pytest.mark.parametrize("shape,n,n_oct,mc", [((48, 64), 5, 3, 8), ((33, 47), 3, 2, 0), ((135, 240), 40, 3, 8), ((70, 130), 70, 2, 8)])
def test_batched_dense_mode_bit_exact(shape, n, n_oct, mc):
    # vslam_params.extrema_dense = 1: the same test inside vslam_detect_batch_dev (grid.z = frames), bitmask and
    # ordered list per frame against the oracle; 40 / 70 frames take the gated / half-batch launch orders
    import torch

    capi.build()
    rows, cols = shape
    frames_np = np.stack([synth.frame_np(rows, cols, f, 3, "noise" if f % 3 == 1 else "checker") for f in range(n)])
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        p = capi.default_params(rows, cols, n_octaves=n_oct, extrema_dense=1, min_contrast=mc, dog_cap=1 << 18)
        L = capi.batch_layout(p)
        dev = "cuda:0"
        o = dict(pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                 extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                 dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
                 dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
        ctx.detect_batch(p, torch.from_numpy(frames_np).to(dev), **o)
        torch.cuda.synchronize()
        bits = o["extrema_bits"].cpu().numpy().view(np.uint64)
        cnt = o["dog_counts"].cpu().numpy()
        for f in sorted(set([0, 1, n // 2 - 1, n // 2, n - 1])):
            want = oracle.Pyramid(frames_np[f], n_oct)
            pts = []
            for oc in range(n_oct):
                wm, wp = want.extrema_dense(oc, mc)
                r, c, wpr = L.lat_rows[oc], L.lat_cols[oc], L.lat_words[oc]
                assert (r, c) == want.sizes[oc]
                w = bits[f][L.bits_offset[oc]: L.bits_offset[oc] + 3 * r * wpr]
                gm = np.unpackbits(w.view(np.uint8).reshape(3, r, wpr * 8), axis=-1, bitorder="little")[..., :c]
                assert gm.tobytes() == wm.tobytes(), ("mask", f, oc)
                pts.append(wp)
            want.close()
            allp = np.concatenate(pts)
            assert cnt[f] == len(allp), (f, int(cnt[f]), len(allp))
            m = min(len(allp), p.dog_cap)
            assert o["dog_points"][f][:m].cpu().numpy().view(capi.POINT_DTYPE).reshape(-1).tobytes() == allp[:m].tobytes(), f
    finally:
        ctx.close()
