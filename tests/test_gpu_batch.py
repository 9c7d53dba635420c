"""Parity of the device-resident batched path (vslam_detect_batch_dev) -- the path bench.py
measures -- against the CPU oracle, frame by frame, plus size-independent properties at the
full BASELINE sizes."""
import numpy as np
import pytest

import oracle
from visualslam_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    capi.build()
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    yield ctx, torch
    ctx.close()


@pytest.fixture(params=["dot", "mx"])
def path(request, env):
    """Both kernel families for the same rows: the default packed-dot kernels and, switched on for the
    shared context only while the test runs, the OPT-IN matrix-core kernels - every parametrised test shows up as a dot of
    its own instead of inside one child-process run."""
    ctx, _ = env
    was = ctx.matrix_path()
    ctx.set_matrix_path(request.param == "mx")
    yield request.param
    ctx.set_matrix_path(was)


def run_batch(ctx, torch, frames_np, with_nms2=True, **pkw):
    n, rows, cols = frames_np.shape
    p = capi.default_params(rows, cols, **pkw)
    L = capi.batch_layout(p)
    dev = "cuda:0"
    frames = torch.from_numpy(frames_np).to(dev)
    o = dict(
        response=torch.empty((n, rows, cols), dtype=torch.float32, device=dev),
        nms_mask=torch.empty((n, rows, cols), dtype=torch.uint8, device=dev),
        nms2=torch.empty((n, rows, cols), dtype=torch.float32, device=dev) if with_nms2 else None,
        harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev),
        harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
        pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
        extrema_bits=torch.zeros((n, max(L.bits_frame_words, 1)), dtype=torch.int64, device=dev),
        dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
        dog_counts=torch.zeros(n, dtype=torch.int32, device=dev),
    )
    if p.orient:
        o["oriented_points"] = torch.zeros((n, p.oriented_cap, 6), dtype=torch.int32, device=dev)
        o["oriented_counts"] = torch.zeros(n, dtype=torch.int32, device=dev)
        o["oriented_survivors"] = torch.full((n,), -1, dtype=torch.int32, device=dev)
        o["descriptors"] = torch.full((n, p.oriented_cap, 128), 7.0, dtype=torch.float32, device=dev)
        o["descriptor_defined"] = torch.full((n, p.oriented_cap), 9, dtype=torch.uint8, device=dev)
    ctx.detect_batch(p, frames, **o)
    torch.cuda.synchronize()
    return p, L, {k: (v.cpu().numpy() if v is not None else None) for k, v in o.items()}


def check_frame(p, L, out, f, img, n_oct):
    R = oracle.harris_response(img)
    assert out["response"][f].tobytes() == R.tobytes()
    assert (out["nms_mask"][f] == oracle.nms_strict(oracle.convert_scale_abs(R), 3)).all()
    n2, _ = oracle.nms2(R, 5)
    if out["nms2"] is not None:
        assert out["nms2"][f].tobytes() == n2.tobytes()
    kps = oracle.harris_keypoints(n2)
    assert out["harris_counts"][f] == len(kps)
    m = min(len(kps), p.harris_cap)
    got = out["harris_kps"][f][:m].copy().view(capi.KP_DTYPE).reshape(-1)
    assert got.tobytes() == kps[:m].tobytes()
    want = oracle.Pyramid(img, n_oct, p.sigma0)
    block = out["pyramid"][f]
    all_pts = []
    for o in range(n_oct):
        r, c = want.sizes[o]
        assert (L.rows[o], L.cols[o]) == (r, c)
        pitch = L.pitch[o]  # plane rows are pitched (cols rounded up to 16); the padding is unspecified
        assert pitch % 16 == 0 and c <= pitch < c + 16
        P = r * pitch
        off = L.octave_offset[o]
        for l in range(6):
            assert (block[off + l * P: off + (l + 1) * P].reshape(r, pitch)[:, :c] == want.gauss(o, l)).all(), ("gauss", o, l)
        for l in range(5):
            assert (block[off + (6 + l) * P: off + (7 + l) * P].reshape(r, pitch)[:, :c] == want.dog(o, l)).all(), ("dog", o, l)
        wm, wp = want.extrema(o, p.extrema_window, p.min_contrast)
        if p.localize:  # list = FeaturePointLocalization survivors
            wp = want.keypoints(o, p.extrema_window)
        lr, lc, wpr = L.lat_rows[o], L.lat_cols[o], L.lat_words[o]
        words = out["extrema_bits"][f][L.bits_offset[o]: L.bits_offset[o] + 3 * lr * wpr].view(np.uint64)
        gm = np.unpackbits(words.view(np.uint8).reshape(3, lr, wpr * 8), axis=-1, bitorder="little")[..., :lc]
        assert (gm == wm).all(), ("mask", o)
        all_pts.append(wp)
    allp = np.concatenate(all_pts)
    assert out["dog_counts"][f] == len(allp)
    m = min(len(allp), p.dog_cap)
    got = out["dog_points"][f][:m].copy().view(capi.POINT_DTYPE).reshape(-1)
    assert got.tobytes() == allp[:m].tobytes()
    if p.orient:  # filterKeypoints per octave, concatenated in octave order
        wo = np.concatenate([want.filter_keypoints(o, all_pts[o]) for o in range(n_oct)])
        # survivors of the edge test = keypoints with at least one oriented point (every survivor has a peak)
        n_surv = len({(int(q["octave"]), int(q["level"]), int(q["row"]), int(q["col"])) for q in wo})
        assert out["oriented_survivors"][f] == n_surv, (f, int(out["oriented_survivors"][f]), n_surv)
        assert n_surv <= p.oriented_cap, "check_frame compares whole lists: raise oriented_cap"
        assert out["oriented_counts"][f] == len(wo), (f, int(out["oriented_counts"][f]), len(wo))
        mo = min(len(wo), p.oriented_cap)
        goo = out["oriented_points"][f][:mo].copy().view(capi.POINT_DTYPE).reshape(-1)
        assert goo.tobytes() == wo[:mo].tobytes()
        # SIFT() on the oriented points, octave by octave, in the batched path
        wd, wk = [], []
        for o in range(n_oct):
            d, k = want.sift_descriptors(o, wo[wo["octave"] == o])
            wd.append(d)
            wk.append(k)
        wd, wk = np.concatenate(wd), np.concatenate(wk)
        assert (out["descriptor_defined"][f][:mo] == wk[:mo]).all()
        assert np.array_equal(out["descriptors"][f][:mo], wd[:mo], equal_nan=True)
        assert (out["descriptors"][f][mo:] == 7.0).all() and (out["descriptor_defined"][f][mo:] == 9).all()  # rows beyond the list untouched
    want.close()


def test_batch_small_frames_all_outputs(env, path):
    ctx, torch = env
    frames = synth.frames_np(11, 96, 160, stream_id=7)
    frames[3] = synth.frame_np(96, 160, kind="noise")
    frames[5] = synth.frame_np(96, 160, kind="constant")
    p, L, out = run_batch(ctx, torch, frames, n_octaves=3)
    for f in range(frames.shape[0]):
        check_frame(p, L, out, f, frames[f], 3)


@pytest.mark.parametrize("mode", ["candidates", "localized", "oriented"])
def test_batch_larger_than_one_chunk(env, mode):
    # more frames than one whole-batch launch takes (256): the second chunk reuses the scratch, the
    # events and the auxiliary streams of the first; frames on both sides of the seam against the oracle
    ctx, torch = env
    frames = synth.frames_np(300, 40, 56, stream_id=13)
    frames[256] = synth.frame_np(40, 56, kind="noise")
    kw = dict(localize=int(mode != "candidates"), orient=int(mode == "oriented"))
    p, L, out = run_batch(ctx, torch, frames, n_octaves=2, harris_cap=1024, dog_cap=2048, **kw)
    for f in (0, 1, 254, 255, 256, 257, 299):
        check_frame(p, L, out, f, frames[f], 2)
    # every frame of the first chunk also appears, identically, nowhere else: spot-check that no
    # frame of chunk 2 picked up results of chunk 1
    assert out["dog_counts"][256] != out["dog_counts"][0] or out["harris_counts"][256] != out["harris_counts"][0]


@pytest.mark.parametrize("lists", [False, True])
def test_two_full_chunks_of_256_frames(env, lists, path):
    # with two chunks of >= 64 frames the second chunk's half-batch upsample runs on a
    # side stream and overwrites the octave bases of frames [128, 256) of the FIRST chunk; in a pyramid-only
    # call nothing but the ev_chunk wait orders it behind the first chunk's octave-0 kernel.  Every frame
    # is different (its own seed; noise frames at the seams), so a pyramid built from the wrong chunk's
    # base cannot pass.  Pyramid-only and list mode; second halves of both chunks against the oracle, and
    # every pyramid of a second run of the same batch equal to the first (a race is not repeatable).
    ctx, torch = env
    rows, cols, n, n_oct = 136, 256, 512, 3  # cols % 32 == 0: pitch == cols at all three octaves (row padding is unspecified)
    frames_np = np.stack([synth.frame_np(rows, cols, frame=f, stream_id=31 + (f >> 8), kind="noise" if f % 64 == 0 else "checker")
                          for f in range(n)])
    dev = "cuda:0"
    frames = torch.from_numpy(frames_np).to(dev)
    p = capi.default_params(rows, cols, n_octaves=n_oct, harris_cap=4096, dog_cap=16384)
    L = capi.batch_layout(p)
    runs = []
    for rep in range(2):
        o = dict(pyramid=torch.zeros((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev))
        if lists:
            o.update(extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                     dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
                     dog_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                     response=torch.empty((n, rows, cols), dtype=torch.float32, device=dev),
                     harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev),
                     harris_counts=torch.zeros(n, dtype=torch.int32, device=dev))
        ctx.detect_batch(p, frames, **o)
        torch.cuda.synchronize()
        runs.append(o)
    assert torch.equal(runs[0]["pyramid"], runs[1]["pyramid"])
    o = runs[0]
    for f in (0, 127, 128, 129, 191, 254, 255, 256, 383, 384, 385, 448, 511):
        want = oracle.Pyramid(frames_np[f], n_oct, p.sigma0)
        blk = o["pyramid"][f].cpu().numpy()
        pts = []
        for oc in range(n_oct):
            r, c = want.sizes[oc]
            pitch, off = L.pitch[oc], L.octave_offset[oc]
            P = r * pitch
            for l in range(6):
                assert (blk[off + l * P: off + (l + 1) * P].reshape(r, pitch)[:, :c] == want.gauss(oc, l)).all(), (f, "gauss", oc, l)
            for l in range(5):
                assert (blk[off + (6 + l) * P: off + (7 + l) * P].reshape(r, pitch)[:, :c] == want.dog(oc, l)).all(), (f, "dog", oc, l)
            pts.append(want.extrema(oc, 3, 8)[1])
        want.close()
        if lists:
            allp = np.concatenate(pts)
            assert int(o["dog_counts"][f]) == len(allp), f
            assert o["dog_points"][f][: len(allp)].cpu().numpy().view(capi.POINT_DTYPE).reshape(-1).tobytes() == allp.tobytes(), f
            kps = oracle.harris_keypoints(oracle.nms2(oracle.harris_response(frames_np[f]), 5)[0])
            assert int(o["harris_counts"][f]) == len(kps), f
            assert o["harris_kps"][f][: len(kps)].cpu().numpy().view(capi.KP_DTYPE).reshape(-1).tobytes() == kps.tobytes(), f


@pytest.mark.parametrize("mode", ["candidates", "localized", "oriented"])
def test_batch_of_40_frames_takes_the_gated_launch_order(env, mode):
    # from 32 frames on the side streams are ordered differently (vslam_hip.hip, enqueue_dog): the Harris
    # chain and the localizing scan wait for the last LDS-tiled octave kernel, the plain scan does not.
    # Same results, every list mode, frames with different content
    ctx, torch = env
    frames = synth.frames_np(40, 96, 160, stream_id=21)
    frames[5] = synth.frame_np(96, 160, kind="noise")
    frames[39] = synth.frame_np(96, 160, frame=3, stream_id=2, kind="noise")
    kw = dict(localize=int(mode != "candidates"), orient=int(mode == "oriented"))
    p, L, out = run_batch(ctx, torch, frames, n_octaves=3, harris_cap=4096, dog_cap=8192, **kw)
    for f in (0, 5, 17, 38, 39):
        check_frame(p, L, out, f, frames[f], 3)


@pytest.mark.parametrize("mode", ["candidates", "localized", "oriented"])
def test_batch_of_70_frames_goes_through_octave_0_in_two_halves(env, mode):
    # from 64 frames on the upsample and octave 0 run as two half-batches (the second half upsampled on
    # the side stream, enqueue_dog); frames on both sides of the seam, first and last, every list mode
    ctx, torch = env
    frames = synth.frames_np(70, 72, 104, stream_id=23)
    frames[34] = synth.frame_np(72, 104, kind="noise")
    frames[35] = synth.frame_np(72, 104, frame=9, stream_id=4, kind="noise")
    kw = dict(localize=int(mode != "candidates"), orient=int(mode == "oriented"))
    p, L, out = run_batch(ctx, torch, frames, n_octaves=3, harris_cap=4096, dog_cap=8192, **kw)
    for f in (0, 33, 34, 35, 36, 69):
        check_frame(p, L, out, f, frames[f], 3)


def test_batch_ragged_size_and_small_caps(env, path):
    ctx, torch = env
    frames = synth.frames_np(2, 75, 131, stream_id=9)
    p, L, out = run_batch(ctx, torch, frames, n_octaves=2, harris_cap=5, dog_cap=7, min_contrast=0)
    for f in range(2):
        check_frame(p, L, out, f, frames[f], 2)


@pytest.mark.parametrize("shape,n_oct,window", [((96, 160), 3, 3), ((75, 131), 2, 3), ((60, 80), 2, 5), ((270, 480), 4, 3)])
def test_batch_localized_keypoints(env, shape, n_oct, window):
    # localize=1: dog_points is initialKeypointDetection's real output (fast and generic kernels)
    ctx, torch = env
    frames = synth.frames_np(3, *shape, stream_id=11)
    frames[1] = synth.frame_np(*shape, kind="noise")
    p, L, out = run_batch(ctx, torch, frames, n_octaves=n_oct, localize=1, extrema_window=window)
    for f in range(3):
        check_frame(p, L, out, f, frames[f], n_oct)


@pytest.mark.parametrize("shape,n_oct", [((96, 160), 3), ((75, 131), 2), ((270, 480), 4), ((40, 56), 3)])
def test_batch_oriented_keypoints(env, shape, n_oct):
    # orient = 1: filterKeypoints inside the batched path (edge test -> survivors -> magnitude region in
    # LDS -> histogram peaks), bit-exact against the oracle; noise frames have thousands of survivors
    ctx, torch = env
    frames = synth.frames_np(4, *shape, stream_id=17)
    frames[1] = synth.frame_np(*shape, kind="noise")
    frames[3] = synth.frame_np(*shape, frame=5, stream_id=3, kind="noise")
    p, L, out = run_batch(ctx, torch, frames, n_octaves=n_oct, localize=1, orient=1)
    assert out["oriented_counts"].sum() > 0
    for f in range(4):
        check_frame(p, L, out, f, frames[f], n_oct)
    # overflow on purpose: with oriented_cap = 8 only the first 8 edge-test survivors are evaluated.  The
    # contract (include/vslam.h): oriented_survivors still reports the true survivor count, which is how
    # the caller sees the truncation; oriented_counts then covers the evaluated survivors only.
    p2, L2, out2 = run_batch(ctx, torch, frames[1:2], n_octaves=n_oct, localize=1, orient=1, oriented_cap=8)
    assert (out2["oriented_points"][0][:8] == out["oriented_points"][1][:8]).all()
    assert out2["oriented_survivors"][0] == out["oriented_survivors"][1] > 8
    full = out["oriented_points"][1][: out["oriented_counts"][1]].copy().view(capi.POINT_DTYPE).reshape(-1)
    first8 = []
    for q in full:  # oriented points of the first 8 distinct keypoints, in list order
        key = (int(q["octave"]), int(q["level"]), int(q["row"]), int(q["col"]))
        if key not in first8:
            if len(first8) == 8:
                break
            first8.append(key)
    n8 = sum((int(q["octave"]), int(q["level"]), int(q["row"]), int(q["col"])) in first8 for q in full)
    assert out2["oriented_counts"][0] == n8 < out["oriented_counts"][1]


def test_config1_640x480_plumbing(env):
    # BASELINE config 1: 640x480 grayscale frame, Harris k=0.04 (synthetic, seed 0x5EED0001)
    ctx, torch = env
    frames = synth.frames_np(1, 480, 640, stream_id=1)
    p, L, out = run_batch(ctx, torch, frames)
    check_frame(p, L, out, 0, frames[0], 4)


def test_config2_and_3_full_1080p_frame(env, path):
    # BASELINE configs 2+3: one 1920x1080 frame, Harris+NMS indices and the 4x(6,5) pyramid +
    # extrema, bit-exact against the oracle
    ctx, torch = env
    frames = synth.frames_np(1, 1080, 1920, stream_id=0)
    p, L, out = run_batch(ctx, torch, frames, with_nms2=False)
    check_frame(p, L, out, 0, frames[0], 4)


def test_full_1080p_frame_localized_and_oriented(env):
    # SURVEY section 8f rows 2-3 at BASELINE's frame size: the fused localize mode of the batch path
    # and the per-image filterKeypoints call against the oracle (noise frame: every stage populated)
    ctx, torch = env
    frames = synth.frame_np(1080, 1920, kind="noise")[None].copy()  # [None] alone leaves stride 0 on the new axis
    p, L, out = run_batch(ctx, torch, frames, with_nms2=False, localize=1, orient=1, oriented_cap=1 << 17)
    check_frame(p, L, out, 0, frames[0], 4)  # includes the batched filterKeypoints list
    want, got = oracle.Pyramid(frames[0], 4, 1.6), ctx.pyramid(frames[0], 4, 1.6)
    n_oriented = 0
    for o in range(4):
        kp = want.keypoints(o, 3)
        gk, nk = got.keypoints(o, 3)
        assert nk == len(kp) and gk.tobytes() == kp.tobytes()
        w = want.filter_keypoints(o, kp)
        g, n = got.filter_keypoints(o, gk)
        assert n == len(w) and g.tobytes() == w.tobytes(), o
        n_oriented += n
    assert n_oriented > 1000
    got.close()
    want.close()


def test_full_size_batch_properties(env):
    # size-independent properties on a 1080p batch (no oracle run per frame):
    #  - identical frames give identical outputs wherever they sit in the batch (chunking,
    #    frame strides); constant frames give zero response / zero DoG / all-candidate masks
    #    (3,672,000 sites) and empty lists
    ctx, torch = env
    n = 10
    base = synth.frames_np(2, 1080, 1920, stream_id=2)
    frames = np.empty((n, 1080, 1920), np.uint8)
    frames[0::3] = base[0]
    frames[1::3] = base[1]
    frames[2::3] = 128
    p, L, out = run_batch(ctx, torch, frames, with_nms2=False)
    valid = L.octave_offset[3] + 11 * L.rows[3] * L.pitch[3]  # the block is rounded up to 256 bytes: the tail is never written
    for f in range(3, n):
        for k in ("response", "nms_mask", "extrema_bits"):
            assert out[k][f].tobytes() == out[k][f % 3].tobytes(), (k, f)
        assert out["pyramid"][f][:valid].tobytes() == out["pyramid"][f % 3][:valid].tobytes(), ("pyramid", f)
        assert out["harris_counts"][f] == out["harris_counts"][f % 3]
        assert out["dog_counts"][f] == out["dog_counts"][f % 3]
        m = out["dog_counts"][f]
        assert out["dog_points"][f][:m].tobytes() == out["dog_points"][f % 3][:m].tobytes()
    c = 2
    assert not out["response"][c].any() and not out["nms_mask"][c].any()
    assert out["harris_counts"][c] == 0 and out["dog_counts"][c] == 0
    sites = 0
    for o in range(4):
        P = L.rows[o] * L.pitch[o]  # 1080p: pitch == cols at every octave
        assert L.pitch[o] == L.cols[o]
        off = L.octave_offset[o]
        assert (out["pyramid"][c][off: off + 6 * P] == 128).all()
        assert not out["pyramid"][c][off + 6 * P: off + 11 * P].any()
        lr, lc, wpr = L.lat_rows[o], L.lat_cols[o], L.lat_words[o]
        w = out["extrema_bits"][c][L.bits_offset[o]: L.bits_offset[o] + 3 * lr * wpr].view(np.uint64)
        sites += int(np.unpackbits(w.view(np.uint8)).sum())
    assert sites == 3_672_000
    # DoG is the saturating difference of adjacent Gaussians, everywhere
    blk = out["pyramid"][0]
    for o in range(4):
        P = L.rows[o] * L.pitch[o]
        off = L.octave_offset[o]
        g = blk[off: off + 6 * P].reshape(6, P).astype(np.int16)
        d = blk[off + 6 * P: off + 11 * P].reshape(5, P)
        assert (d == np.maximum(g[1:] - g[:-1], 0)).all()


def test_batch_argument_errors(env):
    ctx, torch = env
    frames = torch.zeros((1, 32, 32), dtype=torch.uint8, device="cuda:0")
    p = capi.default_params(32, 32)
    with pytest.raises(capi.VslamError):
        ctx.detect_batch(p, frames)  # DoG requested but no pyramid buffer
    bad = capi.default_params(32, 32, extrema_window=4)
    with pytest.raises(capi.VslamError):
        ctx.detect_batch(bad, frames, pyramid=torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0"))


def test_undersized_buffers_are_refused_before_any_launch(env):
    # vslam_batch_out carries the size of every buffer; a buffer smaller than n_frames
    # frames need is VSLAM_ERR_INVALID (-1) and nothing is written
    ctx, torch = env
    rows, cols, n = 64, 96, 3
    frames = torch.from_numpy(synth.frames_np(n, rows, cols)).to("cuda:0")
    p = capi.default_params(rows, cols, n_octaves=2)
    L = capi.batch_layout(p)
    z = capi.batch_out_required(p, n)
    dev = "cuda:0"
    good = dict(response=torch.zeros((n, rows, cols), dtype=torch.float32, device=dev),
                harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev),
                harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                pyramid=torch.zeros(z.pyramid_bytes, dtype=torch.uint8, device=dev),
                extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
                dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
    for k in good:
        bad = dict(good)
        flat = good[k].reshape(-1)
        bad[k] = flat[: flat.numel() - 1].clone()  # one element short
        with pytest.raises(capi.VslamError) as e:
            ctx.detect_batch(p, frames, **bad)
        assert e.value.status == -1 and k in str(e.value), (k, str(e.value))
    torch.cuda.synchronize()
    assert all(not bool(t.any()) for t in good.values())  # the refused calls launched nothing
    ctx.detect_batch(p, frames, **good)  # exactly the required sizes pass
    torch.cuda.synchronize()
    assert bool(good["pyramid"].any())
    # a struct without sizes (struct_size unset) is refused as well
    import ctypes as C
    bo = capi.BatchOut()
    bo.pyramid = good["pyramid"].data_ptr()
    rc = capi.lib().vslam_detect_batch_dev(ctx._h, C.byref(p), frames.data_ptr(), rows * cols, n, C.byref(bo))
    assert rc == -1 and b"struct_size" in capi.lib().vslam_last_error(ctx._h)


@pytest.mark.parametrize("n,cap", [(1, 64), (5, 4096), (300, 700), (513, 96)])
def test_pack_lists(env, n, cap):
    # vslam_pack_lists_dev: the lists of a batch back to back (what a host-fed caller downloads), 12- and
    # 24-byte records, counts above the capacity clipped, a packed buffer that is too small never overrun
    ctx, torch = env
    dev = "cuda:0"
    rng = np.random.default_rng(n * 1000 + cap)
    for k in (3, 6):
        lists = torch.from_numpy(rng.integers(-2**31, 2**31 - 1, size=(n, cap, k), dtype=np.int64).astype(np.int32)).to(dev)
        counts_np = rng.integers(0, cap + cap // 4 + 2, size=n).astype(np.int32)  # some exceed cap
        counts_np[rng.integers(0, n)] = 0
        counts = torch.from_numpy(counts_np).to(dev)
        m = np.minimum(counts_np, cap).astype(np.int64)
        total = int(m.sum())
        packed = torch.full((total * k + 5,), 77, dtype=torch.int32, device=dev)
        offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        ctx.pack_lists(lists, counts, packed, offsets)
        torch.cuda.synchronize()
        off = offsets.cpu().numpy()
        assert (off == np.concatenate([[0], np.cumsum(m)])).all()
        want = np.concatenate([lists[f, : m[f]].cpu().numpy().reshape(-1) for f in range(n)]) if total else np.zeros(0, np.int32)
        got = packed.cpu().numpy()
        assert (got[: total * k] == want).all() and (got[total * k:] == 77).all()
        if total > 3:  # too small a destination: filled up to its end, nothing beyond, offsets unchanged
            small = torch.full((total * k - 2 * k + 1 + 4,), 77, dtype=torch.int32, device=dev)
            view = small[: total * k - 2 * k + 1]
            ctx.pack_lists(lists, counts, view, offsets)
            torch.cuda.synchronize()
            g2 = small.cpu().numpy()
            assert (g2[: view.numel()] == want[: view.numel()]).all() and (g2[view.numel():] == 77).all()
            assert int(offsets[-1]) == total


@pytest.mark.parametrize("n,cap", [(1, 64), (7, 4096), (300, 700)])
def test_pack_points16(env, n, cap):
    # vslam_pack_points16_dev: SLAM::point lists packed back to back as 16-byte records
    # {row, col, value, level | octave << 8 | padding << 16}; vslam_points16_expand on the host gives the lists back byte
    # for byte.  Random records in the ranges the detector writes (and negative values: `value` after localization can be),
    # counts below, at and above the capacity, a packed buffer that is too small.
    ctx, torch = env
    rng = np.random.default_rng(n * 131 + cap)
    dev = "cuda:0"
    pts = np.zeros((n, cap, 6), np.int32)
    pts[..., 0] = rng.integers(0, 1 << 20, (n, cap))
    pts[..., 1] = rng.integers(0, 1 << 20, (n, cap))
    pts[..., 2] = rng.integers(-(1 << 31), (1 << 31) - 1, (n, cap))
    pts[..., 3] = rng.integers(0, 2, (n, cap))
    pts[..., 4] = rng.integers(0, 10, (n, cap))
    pts[..., 5] = rng.integers(0, 6, (n, cap))
    counts = rng.integers(0, cap + 1, n).astype(np.int32)
    counts[0] = cap
    if n > 2:
        counts[1], counts[2] = 0, cap + 17  # an empty frame; a frame that overflowed its capacity
    clipped = np.minimum(counts, cap)
    total = int(clipped.sum())
    lists, cnt = torch.from_numpy(pts).to(dev), torch.from_numpy(counts).to(dev)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    for room in (total + 5, max(total - 3, 0)):
        packed = torch.full((max(room, 1), 4), -1, dtype=torch.int32, device=dev)
        view = packed[:room] if room else packed[:0]
        ctx.pack_points16(lists, cnt, view, offsets)
        torch.cuda.synchronize()
        off = offsets.cpu().numpy()
        assert (off == np.concatenate([[0], np.cumsum(clipped)])).all()
        got = capi.points16_expand(packed.cpu().numpy()[: min(total, room)])
        want = np.concatenate([pts[f, : clipped[f]] for f in range(n)]).reshape(-1, 6)[: min(total, room)]
        assert got.tobytes() == np.ascontiguousarray(want).tobytes()
        if room > total:
            assert (packed[total:] == -1).all()  # nothing written beyond the records
    with pytest.raises(capi.VslamError):  # the records move as 16-byte words: a destination 4 bytes off is refused, nothing is launched
        ctx.pack_points16(lists, cnt, packed.view(-1)[1:], offsets)


@pytest.mark.parametrize("localize", [0, 1])
def test_detect_batch_host_lists(env, localize):
    # vslam_detect_batch_host: numpy frames in, packed lists out (no torch in the call), against the oracle frame by
    # frame; then a Harris-only and a DoG-only call, a destination that is too small, and the argument errors
    ctx, torch = env
    n, rows, cols, n_oct = 5, 150, 217, 3
    frames = synth.frames_np(n, rows, cols, stream_id=4)
    frames[2] = 90  # a frame with nothing in it: empty lists in the middle of the batch
    p = capi.default_params(rows, cols, n_octaves=n_oct, localize=localize)
    res = ctx.detect_batch_host(p, frames)
    hk, hoff, hcnt = res["harris"]
    dp, doff, dcnt = res["dog"]
    assert hcnt[2] == 0 and dcnt[2] == 0 and hoff[0] == 0 and doff[0] == 0
    for f in range(n):
        n2, _ = oracle.nms2(oracle.harris_response(frames[f]), 5)
        kps = oracle.harris_keypoints(n2)
        assert hcnt[f] == len(kps) and hoff[f + 1] - hoff[f] == min(len(kps), p.harris_cap)
        assert hk[int(hoff[f]): int(hoff[f + 1])].tobytes() == kps[: p.harris_cap].tobytes()
        want = oracle.Pyramid(frames[f], n_oct, p.sigma0)
        pts = np.concatenate([want.keypoints(o, p.extrema_window) if localize else want.extrema(o, p.extrema_window, p.min_contrast)[1] for o in range(n_oct)])
        want.close()
        assert dcnt[f] == len(pts) and doff[f + 1] - doff[f] == min(len(pts), p.dog_cap)
        assert dp[int(doff[f]): int(doff[f + 1])].tobytes() == pts[: p.dog_cap].tobytes()
    assert len(hk) == hoff[n] and len(dp) == doff[n] and hoff[n] > 0 and doff[n] > 0
    only_h = ctx.detect_batch_host(p, frames, dog_budget=0)
    assert set(only_h) == {"harris"} and only_h["harris"][0].tobytes() == hk.tobytes()
    only_d = ctx.detect_batch_host(p, frames, harris_budget=0)
    assert set(only_d) == {"dog"} and only_d["dog"][0].tobytes() == dp.tobytes() and (only_d["dog"][1] == doff).all()
    # too small a destination: filled to its end with the first records, offsets and counts still the whole truth
    small = ctx.detect_batch_host(p, frames, harris_budget=int(hoff[n]) - 3, dog_budget=7)
    assert small["harris"][0].tobytes() == hk[:-3].tobytes() and (small["harris"][1] == hoff).all() and (small["harris"][2] == hcnt).all()
    assert small["dog"][0].tobytes() == dp[:7].tobytes() and (small["dog"][1] == doff).all()
    with pytest.raises(capi.VslamError):
        ctx.detect_batch_host(capi.default_params(rows, cols, n_octaves=n_oct, orient=1), frames)
    with pytest.raises(capi.VslamError):
        ctx.detect_batch_host(p, frames, harris_budget=0, dog_budget=0)
    with pytest.raises(ValueError):
        ctx.detect_batch_host(p, frames[:, :-1])


def test_stream_tuner_compares_side_stream_pairs_without_touching_the_results():
    # a context that opted in (vslam_ctx_tune_side_streams) runs its 2nd to 5th full-size batch call on
    # three candidate pairs of side streams and adopts the fastest at the first later call that finds them finished (here
    # the 6th: run_batch synchronises after every call).  Seven identical calls on a fresh context: every call's outputs
    # are byte-identical (the pair never matters for results), the report goes 0 -> 1 -> 2, and a context with the
    # tuner's shape changing under it (another batch size in the middle) simply starts over.  Off by default.
    import torch

    capi.build()
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        frames = synth.frames_np(32, 120, 160, stream_id=5)
        assert ctx.side_stream_report() == (0, 0)
        for _ in range(7):  # not asked for: nothing is compared
            run_batch(ctx, torch, frames)
        assert ctx.side_stream_report() == (0, 0)
    finally:
        ctx.close()
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_side_stream_priority(True)  # the tuner compares pairs of YIELDING side streams (the default since round 5 is the main stream's priority)
        ctx.tune_side_streams(True)
        first = None
        states = []
        for call in range(7):
            _, _, out = run_batch(ctx, torch, frames)
            states.append(ctx.side_stream_report()[1])
            blob = b"".join(out[k].tobytes() for k in ("harris_kps", "harris_counts", "dog_points", "dog_counts", "nms_mask", "extrema_bits"))
            if first is None:
                first = blob
            assert blob == first, call
        assert states == [0, 1, 1, 1, 1, 2, 2], states  # first call: set-up; four timed calls; the sixth decides
        assert 0 <= ctx.side_stream_report()[0] <= 2
    finally:
        ctx.close()
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_side_stream_priority(True)  # the tuner compares pairs of YIELDING side streams (the default since round 5 is the main stream's priority)
        ctx.tune_side_streams(True)
        a, b = synth.frames_np(32, 120, 160, stream_id=6), synth.frames_np(40, 120, 160, stream_id=6)
        small = synth.frames_np(3, 120, 160, stream_id=6)
        for fr in (a, a, small, a, b, b, a, a, small, a, a, a, a):  # small calls run on the pair in use and do not restart anything
            run_batch(ctx, torch, fr)
        assert ctx.side_stream_report()[1] == 2
    finally:
        ctx.close()
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_side_stream_priority(True)  # the tuner compares pairs of YIELDING side streams (the default since round 5 is the main stream's priority)
        ctx.tune_side_streams(True)
        a, b = synth.frames_np(32, 120, 160, stream_id=6), synth.frames_np(40, 120, 160, stream_id=6)
        for fr in (a, a, b, a, b, a, b):  # a caller whose full-size shape keeps changing: given up on the first pair
            run_batch(ctx, torch, fr)
        assert ctx.side_stream_report() == (0, 2)
    finally:
        ctx.close()


def test_tuner_alone_does_not_lock_the_watchdog_out():
    # a caller that used only the documented opt-in - vslam_ctx_tune_side_streams(ctx, 1), no
    # set_side_stream_priority - got neither the tuner nor the watchdog for the life of the context (each waited for the
    # other).  Now the opt-in selects the yielding streams the tuner compares; the comparison ends, then the watch runs and ends.
    import torch

    capi.build()
    rows, cols, n = 540, 960, 48
    dev = "cuda:0"
    frames = synth.frames_torch(n, rows, cols, stream_id=4, device=torch.device(dev))
    p = capi.default_params(rows, cols)
    L = capi.batch_layout(p)

    def outs():
        return dict(response=torch.zeros((n, rows, cols), dtype=torch.float32, device=dev), nms_mask=torch.zeros((n, rows, cols), dtype=torch.uint8, device=dev),
                    harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev), harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                    pyramid=torch.zeros((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                    extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                    dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))

    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        ctx.tune_side_streams(True)  # and nothing else
        assert ctx.join_watch_report()[0] == 0  # asking for the comparison asks for yielding streams
        o = outs()
        for _ in range(40):
            ctx.detect_batch(p, frames, **o)
            torch.cuda.synchronize()
            if ctx.side_stream_report()[1] == 2 and ctx.join_watch_report()[1]:
                break
        assert ctx.side_stream_report()[1] == 2, ctx.side_stream_report()
        lv, done, lag = ctx.join_watch_report()
        assert done and 0.0 <= lag < 1.0, (lv, done, lag)  # the watchdog measured and ended too
        ctx.tune_side_streams(True)  # a finished comparison stays finished: accepted, changes nothing
        assert ctx.side_stream_report()[1] == 2
        ref = {k: v.clone() for k, v in o.items()}
    finally:
        ctx.close()
    #  not after the side streams exist: the watchdog has cached the pair in use by then
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        o = outs()
        ctx.detect_batch(p, frames, **o)
        torch.cuda.synchronize()
        with pytest.raises(capi.VslamError):
            ctx.tune_side_streams(True)
        ctx.tune_side_streams(False)  # switching it off is always possible
    finally:
        ctx.close()
    # a pinned level other than 0: the tuner has nothing to compare, ends at its first call, and the results are the same
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        ctx.pin_side_streams(1)
        ctx.tune_side_streams(True)
        o = outs()
        ctx.detect_batch(p, frames, **o)
        torch.cuda.synchronize()
        assert ctx.side_stream_report() == (0, 2) and ctx.join_watch_report()[:2] == (1, True)
        for k in ("response", "nms_mask", "harris_counts", "pyramid", "extrema_bits", "dog_counts"):
            assert torch.equal(ref[k], o[k]), k
    finally:
        ctx.close()
    # the per-context off switch of the watchdog: no measurement is ever taken
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_join_watch(False)
        o = outs()
        for _ in range(6):
            ctx.detect_batch(p, frames, **o)
            torch.cuda.synchronize()
        assert ctx.join_watch_report() == (1, False, -1.0)
    finally:
        ctx.close()


def test_stream_tuner_never_blocks_the_host():
    # the comparison used to wait on the host (hipEventSynchronize) inside the 6th call of an asynchronous
    # entry point.  Eight calls of ~4 ms of GPU work each are enqueued back to back with no synchronisation: no call's
    # enqueue may take anywhere near one call's GPU time (it decides by hipEventQuery when it finds the timed calls finished)
    import time

    import torch

    capi.build()
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_side_stream_priority(True)  # the tuner compares pairs of YIELDING side streams (the default since round 5 is the main stream's priority)
        ctx.tune_side_streams(True)
        rows, cols, n = 1080, 1920, 64
        p = capi.default_params(rows, cols)
        L = capi.batch_layout(p)
        dev = "cuda:0"
        frames = synth.frames_torch(n, rows, cols, stream_id=2, device=torch.device(dev))
        o = dict(response=torch.empty((n, rows, cols), dtype=torch.float32, device=dev), nms_mask=torch.empty((n, rows, cols), dtype=torch.uint8, device=dev),
                 harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev), harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                 pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                 extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                 dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
        ctx.detect_batch(p, frames, **o)  # first call: workspace, tables
        torch.cuda.synchronize()
        ref = (o["dog_counts"].clone(), o["harris_counts"].clone(), o["extrema_bits"].clone())
        t0 = time.perf_counter()
        enq = []
        for _ in range(8):
            t = time.perf_counter()
            ctx.detect_batch(p, frames, **o)
            enq.append(time.perf_counter() - t)
        torch.cuda.synchronize()
        per_call = (time.perf_counter() - t0) / 8
        # a host-side wait for an earlier call would make the enqueue times add up to the GPU time.  The property is asserted
        # on sums and on the calls that create nothing (calls 2-5 create the candidate streams and their events,
        # runtime calls that can take milliseconds on a busy box - a per-call wall-clock bound on those flakes)
        assert sum(enq) < 0.5 * 8 * per_call, (enq, per_call)
        assert sorted(enq)[len(enq) // 2] < 0.25 * per_call and max(enq[5:]) < 0.5 * per_call, (enq, per_call)
        assert ctx.side_stream_report()[1] in (1, 2)
        ctx.detect_batch(p, frames, **o)  # everything has finished: this call decides
        torch.cuda.synchronize()
        assert ctx.side_stream_report()[1] == 2
        assert torch.equal(o["dog_counts"], ref[0]) and torch.equal(o["harris_counts"], ref[1]) and torch.equal(o["extrema_bits"], ref[2])
    finally:
        ctx.close()


def test_join_watchdog_levels_give_the_same_results_and_the_watch_ends():
    # the library itself notices when its low-priority side streams are held up (the main stream's wait at
    # the end of a call exceeds 3 % of the call) and tries the next form - side streams at the main stream's priority (level 1),
    # then none (level 2) - keeping a form only if it is measurably faster.  Here: (a) a context left alone measures its first full-size calls without blocking,
    # reports a lag fraction and ends the watch (on a quiet box at level 0; whatever level it ends on, results are the
    # same); (b) contexts pinned to levels 0, 1 and 2 (vslam_ctx_pin_side_streams) produce byte-identical outputs.
    import os

    import torch

    capi.build()
    rows, cols, n = 540, 960, 48
    dev = "cuda:0"
    frames = synth.frames_torch(n, rows, cols, stream_id=9, device=torch.device(dev), noise_every=4)
    p = capi.default_params(rows, cols)
    L = capi.batch_layout(p)

    def outs():
        return dict(response=torch.zeros((n, rows, cols), dtype=torch.float32, device=dev), nms_mask=torch.zeros((n, rows, cols), dtype=torch.uint8, device=dev),
                    harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev), harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                    pyramid=torch.zeros((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                    extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                    dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))

    res = {}
    for level in (None, 0, 1, 2):
        ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
        try:
            o = outs()
            if level is None:
                assert ctx.join_watch_report() == (1, False, -1.0)  # the default: side streams at the context stream's priority
                for _ in range(16):  # call 1 of a level is not measured; three more are, and read by a later call once finished
                    ctx.detect_batch(p, frames, **o)
                    torch.cuda.synchronize()
                    if ctx.join_watch_report()[1]:
                        break
                lv, done, lag = ctx.join_watch_report()
                assert done and lv in (0, 1, 2) and 0.0 <= lag < 1.0, (lv, done, lag)
                with pytest.raises(capi.VslamError):  # the side streams exist: their priority can no longer be chosen
                    ctx.set_side_stream_priority(True)
                with pytest.raises(capi.VslamError):
                    ctx.pin_side_streams(2)
            else:
                ctx.pin_side_streams(level)
                assert ctx.join_watch_report()[:2] == (level, True)
                ctx.detect_batch(p, frames, **o)
                torch.cuda.synchronize()
            res[level] = o
        finally:
            ctx.close()
    cnt = res[None]["dog_counts"].cpu().numpy()
    hcn = res[None]["harris_counts"].cpu().numpy()
    assert cnt.min() > 0
    for level in (0, 1, 2):
        for k in ("response", "nms_mask", "harris_counts", "pyramid", "extrema_bits", "dog_counts"):
            assert torch.equal(res[None][k], res[level][k]), (level, k)
        for f in range(n):
            assert torch.equal(res[None]["dog_points"][f, : int(cnt[f])], res[level]["dog_points"][f, : int(cnt[f])]), (level, f)
            assert torch.equal(res[None]["harris_kps"][f, : int(hcn[f])], res[level]["harris_kps"][f, : int(hcn[f])]), (level, f)
    # (c) only calls of one shape are compared: a caller that alternates two full-size shapes never fills a window of three
    # measurements; the watch gives up after a few restarts and stays on the default form
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        o = outs()
        for i in range(12):
            m = n if i % 2 == 0 else n - 8
            ctx.detect_batch(p, frames[:m], **{k: v[:m] for k, v in o.items()})
            torch.cuda.synchronize()
        assert ctx.join_watch_report()[:2] == (1, True)
        for k in ("response", "harris_counts", "pyramid", "dog_counts"):
            assert torch.equal(res[None][k][: n - 8], o[k][: n - 8]), k
    finally:
        ctx.close()


def test_fast_paths_are_the_ones_that_run(env):
    # the specialised kernels must actually be dispatched for the reference configuration
    # (a silent fall-back to the generic kernels would still pass parity)
    ctx, torch = env
    frames = synth.frames_np(1, 480, 640, stream_id=1)
    for name, want in (("k_pyr_octave", 2), ("k_harris_strip", 1), ("k_gauss_h_strip", 2), ("k_resize_linear2x_slide", 1)):
        ctx.kernel_timing_enable(name)
        run_batch(ctx, torch, frames)
        launches, ms = ctx.kernel_timing_read()
        ctx.kernel_timing_enable(None)
        assert launches >= want and ms > 0, (name, launches)
    # widths that are not a multiple of 8 take the same kernels (pitched planes)
    odd = synth.frames_np(1, 310, 438, stream_id=1)
    for name, want in (("k_pyr_octave", 2), ("k_gauss_h_strip", 2)):
        ctx.kernel_timing_enable(name)
        run_batch(ctx, torch, odd)
        launches, ms = ctx.kernel_timing_read()
        ctx.kernel_timing_enable(None)
        assert launches >= want and ms > 0, (name, launches)


@pytest.mark.parametrize("shape,n_oct", [((8, 8), 1), ((24, 40), 2), ((64, 48), 3), ((32, 128), 4)])
def test_batch_tiny_frames(env, shape, n_oct, path):
    # tiny frames through the specialised kernels: halos far larger than the image (repeated
    # reflection), single-tile grids, lattice rows shorter than one ballot word
    ctx, torch = env
    frames = synth.frames_np(3, shape[0], shape[1], stream_id=11)
    frames[1] = synth.frame_np(shape[0], shape[1], kind="noise")
    p, L, out = run_batch(ctx, torch, frames, n_octaves=n_oct)
    for f in range(3):
        check_frame(p, L, out, f, frames[f], n_oct)


def test_serial_stream_mode_matches_oracle():
    # the default run forks the Harris / extrema chains onto auxiliary streams; the single-stream
    # mode (VSLAM_AUX_STREAMS=0 in the diagnostics build lib/libvslam_diag.so, read once per process) must give the same results
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VSLAM_AUX_STREAMS="0", VSLAM_LIBRARY=capi.DIAG_LIB_PATH)  # a diagnostics-build switch (the API: pin_side_streams(2))
    r = subprocess.run([sys.executable, os.path.join(root, "__graft_entry__.py"), "smoke"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout + r.stderr


def test_fused_band_kernel_matches_oracle():
    # the coarse octaves run two strip kernels by default; the fused band kernel (one launch, the row sums
    # stay in LDS) is opt-in (VSLAM_BAND_KERNEL=1, read once per process: measured slower).
    # One child process runs the shape sweep, the ragged / tiny / 1080p cases and the fast-path check under it.
    import os
    import subprocess
    import sys

    if os.environ.get("VSLAM_BAND_KERNEL") == "1":
        pytest.skip("already inside the band-kernel run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VSLAM_BAND_KERNEL="1", VSLAM_LIBRARY=capi.DIAG_LIB_PATH)  # the switch exists in the diagnostics build only
    sel = "random_shapes or ragged or tiny_frames or config2_and_3 or small_frames_all_outputs or band_kernel_is_dispatched"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_batch.py"), "-m", "gpu", "-q", "-x", "-k", sel],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_band_kernel_is_dispatched_when_asked_for(env):
    import os

    if os.environ.get("VSLAM_BAND_KERNEL") != "1":
        pytest.skip("runs inside test_fused_band_kernel_matches_oracle's child process")
    ctx, torch = env
    ctx.kernel_timing_enable("k_gauss_band")
    run_batch(ctx, torch, synth.frames_np(1, 480, 640, stream_id=1))
    launches, ms = ctx.kernel_timing_read()
    ctx.kernel_timing_enable(None)
    assert launches >= 2 and ms > 0


def test_matrix_core_octave_kernel_matches_oracle():
    # OPT-IN path (VSLAM_MX=1 / vslam_ctx_set_matrix_path, never the default: the north star rules MFMA out): octaves 0
    # and 1 of the reference's pyramid as banded matrix products (k_pyr_octave_mx).  One child process runs the shape
    # sweep, the ragged / tiny / 1080p cases, the reference images and the dispatch check under the switch.
    import os
    import subprocess
    import sys

    if os.environ.get("VSLAM_MX") == "1":
        pytest.skip("already inside the matrix-path run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VSLAM_MX="1")
    # (the batched shape sweep - random shapes, ragged, tiny, 1080p, two chunks - runs in-process since round 5: the `path` fixture)
    sel = "matrix_kernel_is_dispatched or batch_on_the_reference_images"
    sel += " or pyramid or golden_fixtures or filter_keypoints or feature_point_localization or process_gradients"  # the per-image API too
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_batch.py"),
                        os.path.join(root, "tests", "test_gpu_ref_images.py"), os.path.join(root, "tests", "test_gpu_parity.py"),
                        "-m", "gpu", "-q", "-x", "-k", sel],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_matrix_path_on_16x16x64_tiles_matches_oracle():
    # round 6: the second MFMA shape of octaves 0-1 (kernels_pyramid_mx16.hip.h: v_mfma_i32_16x16x64_i8, four accumulator registers,
    # <= 128 vector registers, the operand fragments in LDS) - at parity with the 32-wide form, so it lives in the diagnostics build
    # only (VSLAM_MX_FORM=16).  One child process runs every test that is parametrised over the two kernel families, the reference
    # images and the per-image pyramid tests under it: the same oracle comparisons, bit for bit.
    import os
    import subprocess
    import sys

    if os.environ.get("VSLAM_MX_FORM") == "16":
        pytest.skip("already inside the 16-wide run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VSLAM_MX="1", VSLAM_MX_FORM="16", VSLAM_LIBRARY=capi.DIAG_LIB_PATH)
    sel = "mx or matrix_kernel_is_dispatched or batch_on_the_reference_images or pyramid or golden_fixtures or two_chunks"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_batch.py"),
                        os.path.join(root, "tests", "test_gpu_ref_images.py"), os.path.join(root, "tests", "test_gpu_parity.py"),
                        os.path.join(root, "tests", "test_gpu_large.py"), "-m", "gpu", "-q", "-x", "-k", sel],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_matrix_kernel_is_dispatched_when_asked_for(env):
    import os

    if os.environ.get("VSLAM_MX") != "1":
        pytest.skip("runs inside test_matrix_core_octave_kernel_matches_oracle's child process")
    ctx, torch = env
    assert ctx.matrix_path()
    ctx.kernel_timing_enable("k_pyr_octave_mx")
    run_batch(ctx, torch, synth.frames_np(1, 480, 640, stream_id=1))
    launches, ms = ctx.kernel_timing_read()
    ctx.kernel_timing_enable(None)
    assert launches >= 2 and ms > 0


def test_matrix_path_switch_per_context(env):
    # the same batch through the dot kernels and, switched on for this context only, through the matrix-core kernels:
    # every byte of the pyramid and both lists equal
    ctx, torch = env
    frames = synth.frames_np(3, 200, 320, stream_id=5)  # every octave's width a multiple of 16: no row padding, whole blocks comparable
    frames[1] = synth.frame_np(200, 320, kind="noise")
    was = ctx.matrix_path()
    try:
        ctx.set_matrix_path(False)
        p, L, a = run_batch(ctx, torch, frames)
        ctx.set_matrix_path(True)
        ctx.kernel_timing_enable("k_pyr_octave_mx")
        p, L, b = run_batch(ctx, torch, frames)
        launches, _ = ctx.kernel_timing_read()
        ctx.kernel_timing_enable(None)
    finally:
        ctx.set_matrix_path(was)
    assert launches == 4  # all four octaves of the default pyramid have a matrix-core configuration
    for k in a:
        if a[k] is not None:
            assert np.array_equal(a[k], b[k]), k
    for f in range(3):
        check_frame(p, L, b, f, frames[f], 4)


def test_matrix_path_two_chunks_of_1080p_frames(env):
    # on the matrix path the fused lattice scan leaves its site / seam maps in scratch that the octave
    # kernels write on the MAIN stream and k_extrema_pack reads on a low-priority SIDE stream; a call with more than 256
    # frames reuses that scratch for its second chunk, so the second chunk's octave kernels must wait for the first chunk's
    # pack launches (ev_pack).  512 distinct 1080p frames (every eighth one noise), matrix path on: masks, counts and lists of
    # ALL frames against the default path's (device-side compares; that path has no such scratch), and the frames on both
    # sides of the chunk seam against the oracle.
    ctx, torch = env
    n, rows, cols = 512, 1080, 1920
    dev = "cuda:0"
    frames = synth.frames_torch(n, rows, cols, stream_id=77, device=dev, noise_every=8)
    p = capi.default_params(rows, cols)
    L = capi.batch_layout(p)
    pyr = torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev)  # shared by both runs (62 GB)
    was = ctx.matrix_path()
    res = {}
    try:
        for mx in (True, False):
            ctx.set_matrix_path(mx)
            o = dict(pyramid=pyr, extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                     dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
                     dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
            if mx:
                ctx.kernel_timing_enable("k_pyr_octave_mx")
            ctx.detect_batch(p, frames, **o)
            torch.cuda.synchronize()
            if mx:
                launches, _ = ctx.kernel_timing_read()
                ctx.kernel_timing_enable(None)
                assert launches == 8  # four octaves, two chunks
                seam = {f: pyr[f].cpu().numpy() for f in (255, 256)}
            res[mx] = o
    finally:
        ctx.set_matrix_path(was)
    a, b = res[True], res[False]
    assert torch.equal(a["dog_counts"], b["dog_counts"])
    assert torch.equal(a["extrema_bits"], b["extrema_bits"])
    cnt = a["dog_counts"].cpu().numpy()
    assert (cnt <= p.dog_cap).all() and cnt.min() > 0
    for f in range(n):
        assert torch.equal(a["dog_points"][f, : int(cnt[f])], b["dog_points"][f, : int(cnt[f])]), f
    for f in (255, 256):
        img = frames[f].cpu().numpy()
        want = oracle.Pyramid(img, 4, p.sigma0)
        pts = []
        for oc in range(4):
            r, c = want.sizes[oc]
            pitch, off = L.pitch[oc], L.octave_offset[oc]
            P = r * pitch
            for l in range(5):
                assert (seam[f][off + (6 + l) * P: off + (7 + l) * P].reshape(r, pitch)[:, :c] == want.dog(oc, l)).all(), (f, "dog", oc, l)
            wm, wp = want.extrema(oc, 3, p.min_contrast)
            lr, lc, wpr = L.lat_rows[oc], L.lat_cols[oc], L.lat_words[oc]
            words = a["extrema_bits"][f].cpu().numpy()[L.bits_offset[oc]: L.bits_offset[oc] + 3 * lr * wpr].view(np.uint64)
            gm = np.unpackbits(words.view(np.uint8).reshape(3, lr, wpr * 8), axis=-1, bitorder="little")[..., :lc]
            assert (gm == wm).all(), (f, "mask", oc)
            pts.append(wp)
        want.close()
        allp = np.concatenate(pts)
        assert int(cnt[f]) == len(allp)
        m = min(len(allp), p.dog_cap)
        assert a["dog_points"][f][:m].cpu().numpy().view(capi.POINT_DTYPE).reshape(-1).tobytes() == allp[:m].tobytes(), f


@pytest.mark.parametrize("shape,n_oct,mode", [((40, 56), 6, "candidates"), ((75, 131), 5, "localized"), ((17, 200), 6, "candidates"), ((96, 160), 6, "oriented")])
def test_batch_deep_octaves_on_small_frames(env, path, shape, n_oct, mode):
    # the reference's constructor takes ANY octave count, and sigma doubles per octave while the image halves:
    # octave 4 / 5 of these frames are a few pixels across with kernels of 155 .. 977 taps - the strip kernels' wide-kernel
    # reach (2047 taps since round 5; the generic one-thread-per-pixel kernels before) with BORDER_REFLECT_101 folded many
    # times over.  Every plane, mask and list of every octave against the oracle, both kernel families.
    ctx, torch = env
    rows, cols = shape
    frames = synth.frames_np(3, rows, cols, stream_id=rows + cols)
    frames[1] = synth.frame_np(rows, cols, kind="noise")
    p, L, out = run_batch(ctx, torch, frames, n_octaves=n_oct, localize=int(mode != "candidates"), orient=int(mode == "oriented"))
    assert L.rows[n_oct - 1] >= 1 and L.cols[n_oct - 1] >= 1
    for f in range(3):
        check_frame(p, L, out, f, frames[f], n_oct)


def test_batch_random_shapes(env, path):
    # seeded sweep over frame sizes (multiples of 4/8/16 and ragged ones) through every dispatch
    # path of the batch: specialised and generic kernels must agree with the oracle everywhere
    ctx, torch = env
    rng = np.random.default_rng(20261003)
    shapes = set()
    while len(shapes) < 18:
        r = int(rng.integers(2, 150))
        c = int(rng.integers(2, 300))
        if rng.random() < 0.6:
            c = max(8, c // 8 * 8)
        shapes.add((r, c))
    for rows, cols in sorted(shapes):
        n_oct = max(1, min(3, oracle.auto_num_octaves(2 * rows, 2 * cols) + 2))
        while n_oct > 1 and min(rows, cols) * 2 >> (n_oct - 1) < 2:
            n_oct -= 1
        frames = synth.frames_np(2, rows, cols, stream_id=rows * 1000 + cols)
        frames[1] = synth.frame_np(rows, cols, kind="noise")
        mode = int(rng.integers(0, 3))  # plain candidate list / localized / localized + oriented
        p, L, out = run_batch(ctx, torch, frames, n_octaves=n_oct, localize=int(mode >= 1), orient=int(mode == 2))
        for f in range(2):
            check_frame(p, L, out, f, frames[f], n_oct)


def test_results_are_ordered_by_the_stream_alone():
    # the context must launch on the stream the caller works on.  Inputs are produced and
    # counts are read with stream ordering only (no device-wide synchronize before the read): once on
    # torch's default stream (handle 0 -> the legacy NULL stream) and once on a side stream.
    import torch

    capi.build()
    rows, cols, n = 270, 480, 24
    frames_np = synth.frames_np(n, rows, cols, stream_id=21)
    want_h, want_d = [], []
    for f in (0, n - 1):
        R = oracle.harris_response(frames_np[f])
        want_h.append(len(oracle.harris_keypoints(oracle.nms2(R, 5)[0])))
        w = oracle.Pyramid(frames_np[f], 4, 1.6)
        want_d.append(sum(len(w.extrema(o, 3, 8)[1]) for o in range(4)))
        w.close()
    side = torch.cuda.Stream()
    for st in (torch.cuda.default_stream(), side):
        with torch.cuda.stream(st):
            ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
            p = capi.default_params(rows, cols)
            L = capi.batch_layout(p)
            dev = "cuda:0"
            host = torch.from_numpy(frames_np).pin_memory()
            for rep in range(3):
                frames = host.to(dev, non_blocking=True)  # enqueued on `st`, not waited for
                o = dict(
                    response=torch.empty((n, rows, cols), dtype=torch.float32, device=dev),
                    harris_kps=torch.empty((n, p.harris_cap, 3), dtype=torch.int32, device=dev),
                    harris_counts=torch.full((n,), -7, dtype=torch.int32, device=dev),
                    pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                    dog_points=torch.empty((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
                    dog_counts=torch.full((n,), -7, dtype=torch.int32, device=dev),
                )
                ctx.detect_batch(p, frames, **o)
                hc = o["harris_counts"].to("cpu", non_blocking=True)
                dc = o["dog_counts"].to("cpu", non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(st)
                ev.synchronize()  # waits for `st` only
                assert [int(hc[0]), int(hc[-1])] == want_h, (st, rep, hc.tolist())
                assert [int(dc[0]), int(dc[-1])] == want_d, (st, rep, dc.tolist())
            st.synchronize()
            ctx.close()


def test_config4_full_batch_256_frames_1080p(env):
    # BASELINE config 4 at its full size under -m gpu (not only inside bench.py): 256 x 1080p through
    # one vslam_detect_batch_dev call.  No oracle run per frame -- size-independent properties:
    # the batch holds 3 distinct frames (two synthetic streams and a constant one) replicated in an
    # irregular pattern, replicas must agree bit for bit with their first occurrence (lists included),
    # the first occurrences of frame 0 equal the oracle, and the constant frame has every one of its
    # 3,672,000 lattice sites as a candidate and empty lists.
    ctx, torch = env
    n = 256
    base = synth.frames_np(2, 1080, 1920, stream_id=4)
    kind = np.array([(7 * i + i // 5) % 3 for i in range(n)])
    kind[:3] = [0, 1, 2]
    dev = "cuda:0"
    src = torch.from_numpy(np.stack([base[0], base[1], np.full((1080, 1920), 128, np.uint8)])).to(dev)
    frames = src[torch.from_numpy(kind).to(dev)].contiguous()
    p = capi.default_params(1080, 1920)
    L = capi.batch_layout(p)
    o = dict(
        response=torch.empty((n, 1080, 1920), dtype=torch.float32, device=dev),
        nms_mask=torch.empty((n, 1080, 1920), dtype=torch.uint8, device=dev),
        harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev),
        harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
        pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
        extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
        dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
        dog_counts=torch.zeros(n, dtype=torch.int32, device=dev),
    )
    ctx.detect_batch(p, frames, **o)
    torch.cuda.synchronize()
    hc, dc = o["harris_counts"].cpu().numpy(), o["dog_counts"].cpu().numpy()
    assert (hc <= p.harris_cap).all() and (dc <= p.dog_cap).all()
    valid = L.octave_offset[3] + 11 * L.rows[3] * L.pitch[3]  # the block's 256-byte rounding tail is never written
    for k in ("response", "nms_mask", "pyramid", "extrema_bits"):
        for i in range(3, n):  # device-side compares: the 31 GB of pyramids stay in HBM
            a, b = (o[k][i][:valid], o[k][int(kind[i])][:valid]) if k == "pyramid" else (o[k][i], o[k][int(kind[i])])
            assert torch.equal(a, b), (k, i)  # frames 0, 1, 2 are the first occurrences
    assert (hc == hc[kind]).all() and (dc == dc[kind]).all()
    for k, cnt in (("harris_kps", hc), ("dog_points", dc)):
        for i in (3, 100, 255):
            m = int(cnt[i])
            assert bool((o[k][i][:m] == o[k][int(kind[i])][:m]).all()), (k, i)
    # frame 0 against the oracle: lists and counts (images are covered by the one-frame tests)
    R = oracle.harris_response(base[0])
    kps = oracle.harris_keypoints(oracle.nms2(R, 5)[0])
    assert hc[0] == len(kps)
    assert o["harris_kps"][0][: len(kps)].cpu().numpy().view(capi.KP_DTYPE).reshape(-1).tobytes() == kps.tobytes()
    want = oracle.Pyramid(base[0], 4, 1.6)
    allp = np.concatenate([want.extrema(oc, 3, 8)[1] for oc in range(4)])
    want.close()
    assert dc[0] == len(allp)
    assert o["dog_points"][0][: len(allp)].cpu().numpy().view(capi.POINT_DTYPE).reshape(-1).tobytes() == allp.tobytes()
    # the constant frame
    assert hc[2] == 0 and dc[2] == 0 and not bool(o["response"][2].any()) and not bool(o["nms_mask"][2].any())
    bits = o["extrema_bits"][2].cpu().numpy().view(np.uint64)
    assert int(np.unpackbits(bits.view(np.uint8)).sum()) == 3_672_000


def test_orientation_packed_and_scalar_kernels_agree(env, tmp_path):
    # The fine octaves' orientation histograms run in k_orient_survivors_pk (packed f32, one launch per level); the round-3
    # kernel k_orient_survivors still serves the coarse octaves and, under VSLAM_ORIENT_SCALAR=1, all of them.  Same frames
    # through both (the second in a child process: the switch is read when a context is created): interior survivors
    # (patch path), border survivors (pixel-by-pixel path) and all three levels' spans, byte for byte.
    import os
    import subprocess
    import sys

    ctx, torch = env
    frames = np.stack([synth.frame_np(360, 520, 0, 3, "noise"), synth.frame_np(360, 520, 1, 4, "checker"), synth.frame_np(360, 520, 2, 5, "noise")])
    np.save(tmp_path / "frames.npy", frames)
    p, L, a = run_batch(ctx, torch, frames, n_octaves=3, localize=1, orient=1)
    assert int(a["oriented_counts"].sum()) > 5000
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from tests.test_gpu_batch import run_batch\n"
            "from visualslam_amd import capi\n"
            "capi.build(); ctx = capi.Context(0)\n"
            "p, L, b = run_batch(ctx, torch, np.load(%r), n_octaves=3, localize=1, orient=1)\n"
            "np.savez(%r, pts=b['oriented_points'], cnt=b['oriented_counts'])\n") % (root, str(tmp_path / "frames.npy"), str(tmp_path / "scalar.npz"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VSLAM_ORIENT_SCALAR="1", VSLAM_LIBRARY=capi.DIAG_LIB_PATH), cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    b = np.load(tmp_path / "scalar.npz")
    assert np.array_equal(a["oriented_counts"], b["cnt"])
    for f in range(len(frames)):
        n = int(min(a["oriented_counts"][f], p.oriented_cap))
        assert np.array_equal(a["oriented_points"][f][:n], b["pts"][f][:n]), f
    for f in range(len(frames)):
        check_frame(p, L, a, f, frames[f], 3)


@pytest.mark.parametrize("sigma0", [0.9, 1.3, 2.0])
def test_batch_oriented_keypoints_other_base_sigmas(env, sigma0):
    # The orientation blur's width follows sigma0 (1.5 * sigma(octave, level), 8 sigma + 1 taps): other base sigmas give the
    # packed kernel other spans, column-group counts and LDS layouts per level (sigma0 = 0.9: spans 30-38; 1.3: 36-48) and,
    # at 2.0 (spans 46, 54, 64: the widest is past the packed kernel's 60), hand octave 0 back to the round-3 kernel.
    ctx, torch = env
    frames = np.stack([synth.frame_np(300, 400, 0, 7, "noise"), synth.frame_np(300, 400, 1, 8, "checker")])
    p, L, out = run_batch(ctx, torch, frames, n_octaves=2, localize=1, orient=1, sigma0=sigma0)
    assert int(out["oriented_counts"].sum()) > 300
    for f in range(2):
        check_frame(p, L, out, f, frames[f], 2)


@pytest.mark.parametrize("n,mode,mx", [(8, {}, False), (8, dict(localize=1, orient=1, describe=1), False), (64, dict(localize=1, orient=1), False), (64, {}, True)])
def test_batch_call_captured_into_a_graph(env, n, mode, mx):
    # vslam_detect_batch_dev inside a stream capture (hipGraph), after one warm-up call with the same parameters (workspace,
    # taps and side streams exist then: a capture can allocate nothing): the side-stream forks are all capturable, the stream
    # tuner stays out, and the two nested forks of the orientation stage fall back to their own stream (this runtime faults
    # at the end of a capture in which a side stream forks to another one).  Replays on new frame contents = eager calls.
    _, torch = env
    dev = "cuda:0"
    st = torch.cuda.Stream()
    rows, cols = (240, 320) if n == 8 else (120, 160)
    mode = dict(mode)
    describe = mode.pop("describe", 0)
    with torch.cuda.stream(st):
        ctx = capi.Context(0, st.cuda_stream)
        ctx.set_matrix_path(mx)
        p = capi.default_params(rows, cols, n_octaves=3, **mode)
        L = capi.batch_layout(p)
        frames = torch.from_numpy(synth.frames_np(n, rows, cols, stream_id=3)).to(dev)

        def outs():
            o = dict(response=torch.zeros((n, rows, cols), dtype=torch.float32, device=dev), nms_mask=torch.zeros((n, rows, cols), dtype=torch.uint8, device=dev),
                     harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev), harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                     pyramid=torch.zeros((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                     extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                     dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
            if mode:
                o.update(oriented_points=torch.zeros((n, p.oriented_cap, 6), dtype=torch.int32, device=dev), oriented_counts=torch.zeros(n, dtype=torch.int32, device=dev))
            if describe:
                o.update(descriptors=torch.zeros((n, p.oriented_cap, 128), dtype=torch.float32, device=dev),
                         descriptor_defined=torch.zeros((n, p.oriented_cap), dtype=torch.uint8, device=dev))
            return o

        a, b = outs(), outs()
        ctx.detect_batch(p, frames, **a)  # warm-up
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            ctx.detect_batch(p, frames, **b)
        lists = {"harris_kps": "harris_counts", "dog_points": "dog_counts", "oriented_points": "oriented_counts", "descriptors": "oriented_counts",
                 "descriptor_defined": "oriented_counts"}
        for rep in range(2):
            kind = "noise" if rep else "checker"
            frames.copy_(torch.from_numpy(np.stack([synth.frame_np(rows, cols, f, 20 + rep, kind) for f in range(n)])).to(dev))
            ctx.detect_batch(p, frames, **a)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert int(a["dog_counts"].sum()) > 0
            for k in a:
                if k in lists:  # records past a frame's count are whatever an earlier call left there
                    for f in range(n):
                        m = int(min(a[lists[k]][f], a[k].shape[1]))
                        assert torch.equal(a[k][f][:m], b[k][f][:m]), (k, f)
                else:
                    assert torch.equal(a[k], b[k]), k
        del g
        ctx.close()


def test_two_chunks_through_the_orientation_and_descriptor_stages(env):
    # 512 frames = two chunks of 256 inside one call: the second chunk's orientation stage re-records the events of the first
    # one's spread launches and early edge test, and queues behind them on the side streams.  The call's oriented points and
    # descriptors = those of two calls of 256 frames each (single-chunk calls are what the oracle tests cover).
    ctx, torch = env
    rows, cols, n, n_oct = 136, 256, 512, 2
    frames_np = np.stack([synth.frame_np(rows, cols, frame=f, stream_id=41 + (f >> 8), kind="noise" if f % 8 == 0 else "checker") for f in range(n)])
    dev = "cuda:0"
    frames = torch.from_numpy(frames_np).to(dev)
    p = capi.default_params(rows, cols, n_octaves=n_oct, localize=1, orient=1, harris_cap=4096, dog_cap=16384)
    L = capi.batch_layout(p)

    def outs(m):
        return dict(pyramid=torch.zeros((m, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                    extrema_bits=torch.zeros((m, L.bits_frame_words), dtype=torch.int64, device=dev),
                    dog_points=torch.zeros((m, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(m, dtype=torch.int32, device=dev),
                    oriented_points=torch.zeros((m, p.oriented_cap, 6), dtype=torch.int32, device=dev), oriented_counts=torch.zeros(m, dtype=torch.int32, device=dev),
                    descriptors=torch.zeros((m, p.oriented_cap, 128), dtype=torch.float32, device=dev),
                    descriptor_defined=torch.zeros((m, p.oriented_cap), dtype=torch.uint8, device=dev))

    whole = outs(n)
    ctx.detect_batch(p, frames, **whole)
    torch.cuda.synchronize()
    assert int(whole["oriented_counts"].sum()) > 10000
    for half in range(2):
        part = outs(256)
        ctx.detect_batch(p, frames[256 * half:256 * (half + 1)], **part)
        torch.cuda.synchronize()
        sl = slice(256 * half, 256 * (half + 1))
        assert torch.equal(whole["dog_counts"][sl], part["dog_counts"]) and torch.equal(whole["oriented_counts"][sl], part["oriented_counts"])
        for f in range(0, 256, 8):  # the noise frames carry the points
            m = int(min(part["oriented_counts"][f], p.oriented_cap))
            assert torch.equal(whole["oriented_points"][256 * half + f][:m], part["oriented_points"][f][:m]), (half, f)
            d = part["descriptor_defined"][f][:m].bool()
            assert torch.equal(whole["descriptor_defined"][256 * half + f][:m], part["descriptor_defined"][f][:m])
            assert torch.equal(whole["descriptors"][256 * half + f][:m][d], part["descriptors"][f][:m][d]), (half, f)
        del part
