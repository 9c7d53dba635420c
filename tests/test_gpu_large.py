"""Parity at the sizes the rest of the suite does not reach: 4K frames with the automatic octave count
(GaussPyramid.cpp:150-152: seven octaves at 3840 x 2160, the coarsest through the generic kernels), a five-octave
ragged frame, a portrait frame - per image (pyramid, localized keypoints, filterKeypoints, Harris response) - and one
BATCHED 4K case through the fused entry point, on the default path and on the opt-in matrix path."""
import numpy as np
import pytest

import oracle
from visualslam_amd import capi, synth

from tests.test_gpu_batch import check_frame, run_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,n_oct", [((2160, 3840), 0), ((1234, 2050), 5), ((3000, 500), 5)])
def test_per_image_api_at_large_sizes(shape, n_oct):
    ctx = capi.Context(0)
    try:
        img = synth.frame_np(*shape, kind="checker")
        got = ctx.pyramid(img, n_oct, 1.6)
        want = oracle.Pyramid(img, n_oct or oracle.auto_num_octaves(*shape), 1.6)
        try:
            assert got.n_octaves == want.n_octaves and (n_oct or got.n_octaves == 7)  # floor(log2(2160)) - 4
            for o in range(got.n_octaves):
                for l in range(6):
                    assert (got.gauss(o, l) == want.gauss(o, l)).all(), ("gauss", o, l)
                for l in range(5):
                    assert (got.dog(o, l) == want.dog(o, l)).all(), ("dog", o, l)
                wk = want.keypoints(o, 3)
                gk, n = got.keypoints(o, 3)
                assert n == len(wk) and gk.tobytes() == wk.tobytes(), ("keypoints", o)
                wf = want.filter_keypoints(o, wk)
                gf, nf = got.filter_keypoints(o, gk)
                assert nf == len(wf) and gf.tobytes() == wf.tobytes(), ("oriented", o)
        finally:
            got.close()
            want.close()
        assert ctx.harris_response(img).tobytes() == oracle.harris_response(img).tobytes()
    finally:
        ctx.close()


@pytest.mark.parametrize("matrix_path", [False, True])
@pytest.mark.parametrize("localize", [0, 1])
def test_batched_4k_frames(matrix_path, localize):
    import torch

    rows, cols = 2160, 3840
    n_oct = oracle.auto_num_octaves(rows, cols)
    assert n_oct == 7  # floor(log2(2160)) - 4 (GaussPyramid.cpp:151)
    frames = synth.frames_np(2, rows, cols, stream_id=3)
    frames[1, :, : cols // 2] = synth.frame_np(rows, cols, kind="noise")[:, : cols // 2]  # half checkerboard stream, half noise
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_matrix_path(matrix_path)
        p, L, out = run_batch(ctx, torch, frames, with_nms2=False, n_octaves=n_oct, localize=localize)
        assert L.rows[0] == 2 * rows and L.cols[0] == 2 * cols
        for f in range(2):
            check_frame(p, L, out, f, frames[f], n_oct)
    finally:
        ctx.close()
        torch.cuda.empty_cache()
