"""CPU: the restated average_if_close / merge_n core (oracle/pyoracle.py, s2p/fusion.py:16-68) against the fixture
produced with the reference's own average_if_close (tests/golden/fusion_stack.npz, made by make_golden.py)."""
import numpy as np

from helpers import load_golden, same


def test_oracle_merge_matches_the_reference_function(oracle):
    g = load_golden("fusion_stack")
    for k in range(4):
        st = list(g["stack%d" % k])
        out = oracle.oracle_merge_n(st, list(g["offsets%d" % k]), "average_if_close", float(g["threshold%d" % k]))
        assert same(out, g["expected%d" % k]), k
        assert np.isfinite(out).any() and np.isnan(out).any()


def test_average_if_close_cases(oracle):
    f = oracle.average_if_close
    assert f(np.array([1.0, 2.0, np.nan]), 3) == 1.5
    assert np.isnan(f(np.array([1.0, 5.0, np.nan]), 3))
    assert f(np.array([np.nan, 4.0, np.nan]), 0) == 4.0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.isnan(f(np.array([np.nan, np.nan]), 1))
