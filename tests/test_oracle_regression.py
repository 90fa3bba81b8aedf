"""The oracle against its own frozen outputs (tests/golden/oracle_regression.npz — a regression pin, NOT a reference vector:
see tests/golden/README.md)."""
import os

import numpy as np


def test_oracle_outputs_have_not_drifted():
    from tests.golden.make_oracle_regression import cases
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_regression.npz"))
    got = cases()
    assert set(ref.files) == set(got)
    for key in ref.files:
        a, b = got[key], ref[key]
        assert a.shape == b.shape, key
        assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, float(np.max(np.abs(b)))), key
