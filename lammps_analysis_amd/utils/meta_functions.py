"""Small host helpers (subset of mdsuite/utils/meta_functions.py of the reference
that the hot path uses: :54-93, :96-129, :132-158)."""
from __future__ import annotations

import json
import os

import numpy as np
import psutil


def is_jsonable(x) -> bool:
    try:
        json.dumps(x)
        return True
    except (TypeError, OverflowError):
        return False


def join_path(a, b) -> str:
    """Join two dataset path components with forward slashes."""
    return os.path.join(a, b).replace("\\", "/")


def get_dimensionality(box) -> int:
    zeros = sum(1 for b in box[:3] if b == 0)
    return 3 - zeros if zeros < 3 else 1


def get_machine_properties() -> dict:
    """Host memory and core count.  ``memory`` is the *available* RAM in bytes,
    which the memory manager turns into a batch size (meta_functions.py:132-158);
    GPU memory is not used for batching, exactly as in the reference, where the
    GPU branch needs more than one TensorFlow GPU to be taken (:49-51)."""
    return {
        "cpu": psutil.cpu_count(logical=True),
        "memory": psutil.virtual_memory().available,
        "gpu": {},
    }


def apply_savgol_filter(data: np.ndarray, order: int = 2, window_length: int = 17) -> np.ndarray:
    """SciPy Savitzky-Golay smoothing with the reference's defaults
    (mdsuite/utils/meta_functions.py:327-355)."""
    from scipy.signal import savgol_filter
    return savgol_filter(data, window_length, order)


def closest_point(data: np.ndarray, value: float):
    """The entry of ``data`` closest to ``value`` (meta_functions.py:358-373)."""
    data = np.asarray(data)
    return data[np.abs(data - value).argmin()]


def golden_section_search(data, a: float, b: float, tol: float = 1e-5, h: float = None,
                          c: float = None, d: float = None, fc: float = None,
                          fd: float = None) -> tuple:
    """Golden-section search for the minimum of tabulated data ``[x, y]`` between ``a``
    and ``b``, probing at the grid points closest to the section points; returns the
    interval that still brackets the minimum (meta_functions.py:376-437)."""
    from .units import golden_ratio
    phi_a = 1 / golden_ratio
    phi_b = 1 / (golden_ratio**2)
    (a, b) = (min(a, b), max(a, b))
    if h is None:
        h = b - a
    if h <= tol:
        return a, b
    if c is None:
        c = closest_point(data[0], a + phi_b * h)
    if d is None:
        d = closest_point(data[0], a + phi_a * h)
    if fc is None:
        fc = data[1][np.where(data[0] == c)]
    if fd is None:
        fd = data[1][np.where(data[0] == d)]
    if fc < fd:
        return golden_section_search(data, a, d, tol, h * phi_a, c=None, fc=None, d=c, fd=fc)
    return golden_section_search(data, c, b, tol, h * phi_a, c=d, fc=fd, d=None, fd=None)
