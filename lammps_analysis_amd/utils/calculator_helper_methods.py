"""Host-side post-processing shared by the Einstein calculators.

``fit_einstein_curve`` follows mdsuite/utils/calculator_helper_methods.py:41-107 of
the reference: it is O(data_range^2) SciPy work on a short series and stays on
the host.  The pairwise ``correlate`` / ``msd_operation`` helpers of the same
file (:110-184) belong to the distinct-diffusion calculators.
"""
from __future__ import annotations

import numpy as np


def _line(x, m, a):
    return m * x + a


def cumulative_trapezoid(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """``scipy.integrate.cumulative_trapezoid(y, x=x)`` for one-dimensional series, in
    SciPy's own operation order (``cumsum(diff(x) * (y[1:] + y[:-1]) / 2.0)``), so that the
    Green-Kubo calculators need no SciPy import for a running integral."""
    y = np.asarray(y, dtype=float)
    x = np.asarray(x, dtype=float)
    return np.cumsum(np.diff(x) * (y[1:] + y[:-1]) / 2.0)


def fit_einstein_curve(x_data: np.ndarray, y_data: np.ndarray, fit_max_index: int):
    """Growing-window linear fits of an MSD curve.

    The fit starts where the second derivative of a quartic interpolating spline
    is closest to zero (values below 1e-5 count as zero), and a straight line is
    fitted to ``[start, i)`` for every ``i`` from ``start + 2`` on.  Returns the
    fit at ``i == fit_max_index`` plus all slopes and their standard errors.
    """
    # SciPy's interpolation and optimisation packages take a few tenths of a second to
    # import; only the fits need them
    from scipy.interpolate import UnivariateSpline
    from scipy.optimize import curve_fit

    x_data = np.asarray(x_data, dtype=float)
    y_data = np.asarray(y_data, dtype=float)
    second = UnivariateSpline(x_data, y_data, s=0, k=4).derivative(n=2)(x_data)
    second[np.abs(second) < 1e-5] = 0
    start = int(np.argmin(np.abs(second)))

    popt, pcov = [], []
    gradients, gradient_errors = [], []
    for stop in range(start + 2, len(y_data)):
        params, cov = curve_fit(_line, xdata=x_data[start:stop], ydata=y_data[start:stop])
        gradients.append(params[0])
        gradient_errors.append(np.sqrt(np.diag(cov))[0])
        if stop == fit_max_index:
            popt, pcov = params, cov
    return popt, pcov, gradients, gradient_errors


def correlate(ds_a: np.ndarray, ds_b: np.ndarray) -> np.ndarray:
    """Full cross-correlation of two ``[T, dims]`` series averaged over dims, lags
    ``0..T-1`` (``jnp.correlate(a, b, "full")`` second half in the reference,
    :110-150).  Host NumPy version used for small inputs and tests."""
    ds_a = np.asarray(ds_a, dtype=float)
    ds_b = np.asarray(ds_b, dtype=float)
    per_dim = [np.correlate(ds_a[:, d], ds_b[:, d], mode="full") for d in range(ds_a.shape[1])]
    acf = np.mean(per_dim, axis=0)
    return acf[int(len(acf) / 2):]


def msd_operation(ds_a: np.ndarray, ds_b: np.ndarray) -> np.ndarray:
    """``mean_dims (a - a[0]) * (b - b[0])`` (:153-184 of the reference)."""
    ds_a = np.asarray(ds_a, dtype=float)
    ds_b = np.asarray(ds_b, dtype=float)
    return np.mean((ds_a - ds_a[0]) * (ds_b - ds_b[0]), axis=-1)
