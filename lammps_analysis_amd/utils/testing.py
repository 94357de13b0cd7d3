"""Recursive almost-equal assertion (mdsuite/utils/testing.py:34-62 semantics)."""
import numpy as np


def assertDeepAlmostEqual(expected, actual, *args, **kwargs):
    if isinstance(expected, dict):
        assert set(expected) == set(actual)
        for key in expected:
            assertDeepAlmostEqual(expected[key], actual[key], *args, **kwargs)
    elif isinstance(expected, (list, tuple)) and expected and isinstance(expected[0], (dict, list, tuple)):
        assert len(expected) == len(actual)
        for e, a in zip(expected, actual):
            assertDeepAlmostEqual(e, a, *args, **kwargs)
    else:
        np.testing.assert_array_almost_equal(np.asarray(expected), np.asarray(actual),
                                             *args, **kwargs)
