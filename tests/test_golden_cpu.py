"""The oracle reproduces the committed golden vectors (tests/golden/*.npz)."""
import os

import numpy as np
import pytest

import oracle
from tests.golden.make_golden import CASES, make

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    want = np.load(os.path.join(GOLD, name + ".npz"))
    got = make(name)
    assert set(want.files) == set(got)
    for k in want.files:
        w, g = want[k], np.asarray(got[k])
        assert w.dtype == g.dtype and w.shape == g.shape, k
        assert w.tobytes() == g.tobytes(), k
