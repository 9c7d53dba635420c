"""The dense 3x3x3 scale-space test (an extension, SURVEY.md section 8a note: the north star's wording;
parity is judged on the reference's stride-3 lattice test, not on this).  CPU side: the oracle's loop
against scipy's separable rank filters on the same DoG stack."""
import numpy as np
import pytest
from scipy import ndimage

import oracle
from visualslam_amd import synth

from tests import refimg


def scipy_dense(pyr, o, mc):
    D = np.stack([pyr.dog(o, l) for l in range(5)]).astype(np.int32)
    mx = ndimage.maximum_filter(D, size=3, mode="nearest")
    mn = ndimage.minimum_filter(D, size=3, mode="nearest")
    mask = ((D == mx) | (D == mn))[1:4]
    lv, ys, xs = np.nonzero(mask & (D[1:4] >= mc))
    return mask.astype(np.uint8), lv + 1, ys, xs, D[1:4][lv, ys, xs]


@pytest.mark.parametrize("shape,kind", [((70, 93), "noise"), ((48, 64), "checker"), ((1, 1), "noise"), ((2, 130), "noise"), ((67, 3), "checker")])
@pytest.mark.parametrize("mc", [8, 0])
def test_oracle_dense_extrema_equals_rank_filters(shape, kind, mc):
    img = synth.frame_np(shape[0], shape[1], 0, 3, kind)
    n_oct = 2 if min(shape) >= 8 else 1
    p = oracle.Pyramid(img, n_oct)
    try:
        for o in range(n_oct):
            mask, pts = p.extrema_dense(o, mc)
            wm, lv, ys, xs, vals = scipy_dense(p, o, mc)
            assert mask.shape == wm.shape and (mask == wm).all()
            assert len(pts) == len(lv)
            assert (pts["level"] == lv).all() and (pts["row"] == ys + 1).all() and (pts["col"] == xs + 1).all()
            assert (pts["value"] == vals).all() and (pts["padding"] == 1).all() and (pts["octave"] == o).all()
    finally:
        p.close()


def test_dense_extrema_contains_the_lattice_candidates_it_should():
    # a site of the reference's stride-3 lattice that is an extremum of the FULL 3x3x3 neighbourhood is
    # also an extremum of the reference's 2x2x3 sub-window: dense candidates on lattice sites are a
    # subset of the lattice candidates
    img = refimg.load("blox")
    p = oracle.Pyramid(img, 2)
    try:
        for o in range(2):
            dense, _ = p.extrema_dense(o, 0)
            lat, _ = p.extrema(o, 3, 0)
            sub = dense[:, 0::3, 0::3][:, : lat.shape[1], : lat.shape[2]]  # padded (1+3li, 1+3lj) = unpadded (3li, 3lj)
            assert sub.shape == lat.shape and not (sub & ~lat.astype(bool)).any()
            assert 0 < sub.sum() <= lat.sum()
    finally:
        p.close()
