"""ctypes access to oracle/liboracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY; see oracle/mdx_oracle.c for what each function restates.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import numpy as np

from . import build as _build

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = _build.LIB
        if not path.exists():
            path = _build.build_oracle()
        _lib = ctypes.CDLL(str(path))
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def rdf_hist(positions: np.ndarray, n_per_species: Sequence[int], box, cutoff, nbins: int,
             frames=None, strict: bool = True) -> np.ndarray:
    """Counts ``[n_keys, nbins]`` (uint64) of the species-concatenated positions
    ``[N, F, 3]`` float32."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    nps = np.ascontiguousarray(n_per_species, dtype=np.int64)
    assert pos.ndim == 3 and pos.shape[2] == 3 and pos.shape[0] == int(nps.sum())
    n_species = len(nps)
    n_keys = n_species * (n_species + 1) // 2
    if frames is None:
        frames = np.arange(pos.shape[1], dtype=np.int64)
    frames = np.ascontiguousarray(frames, dtype=np.int64)
    boxf = np.ascontiguousarray(box, dtype=np.float32)
    hist = np.zeros((n_keys, nbins), dtype=np.uint64)
    rc = lib().oracle_rdf_hist(
        _ptr(pos, ctypes.c_float), ctypes.c_int(n_species), _ptr(nps, ctypes.c_int64),
        ctypes.c_int64(pos.shape[1]), _ptr(frames, ctypes.c_int64),
        ctypes.c_int64(len(frames)), _ptr(boxf, ctypes.c_float),
        ctypes.c_float(np.float32(cutoff)), ctypes.c_int(nbins),
        ctypes.c_int(1 if strict else 0), _ptr(hist, ctypes.c_uint64))
    if rc != 0:
        raise RuntimeError(f"oracle_rdf_hist failed with {rc}")
    return hist


def weighted_atom_sum(x: np.ndarray, weight: float) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_atoms, n_frames, k = x.shape
    out = np.zeros((n_frames, k), dtype=np.float64)
    lib().oracle_weighted_atom_sum_f32(
        _ptr(x, ctypes.c_float), ctypes.c_int64(n_atoms), ctypes.c_int64(n_frames),
        ctypes.c_int(k), ctypes.c_double(weight), _ptr(out, ctypes.c_double),
        ctypes.c_int(0))
    return out


def acf_windows(x: np.ndarray, correlation_time: int, win_begin: int, win_end: int,
                tau: np.ndarray, scale: float = 1.0, per_window: bool = False):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n_series, n_frames, k = x.shape
    tau = np.ascontiguousarray(tau, dtype=np.int32)
    R = len(tau)
    acf = np.zeros(R, dtype=np.float64)
    pw = np.zeros((max(win_end - win_begin, 1), R), dtype=np.float64) if per_window else None
    lib().oracle_acf_windows(
        _ptr(x, ctypes.c_double), ctypes.c_int64(n_series), ctypes.c_int64(n_frames),
        ctypes.c_int(k), ctypes.c_int64(correlation_time), ctypes.c_int64(win_begin),
        ctypes.c_int64(win_end), _ptr(tau, ctypes.c_int32), ctypes.c_int(R),
        ctypes.c_double(scale), _ptr(acf, ctypes.c_double),
        _ptr(pw, ctypes.c_double) if per_window else None)
    return (acf, pw) if per_window else acf


def msd_windows(x: np.ndarray, correlation_time: int, win_begin: int, win_end: int,
                tau: np.ndarray, scale: float = 1.0) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float64)
    n_series, n_frames, k = x.shape
    tau = np.ascontiguousarray(tau, dtype=np.int32)
    R = len(tau)
    msd = np.zeros(R, dtype=np.float64)
    lib().oracle_msd_windows(
        _ptr(x, ctypes.c_double), ctypes.c_int64(n_series), ctypes.c_int64(n_frames),
        ctypes.c_int(k), ctypes.c_int64(correlation_time), ctypes.c_int64(win_begin),
        ctypes.c_int64(win_end), _ptr(tau, ctypes.c_int32), ctypes.c_int(R),
        ctypes.c_double(scale), _ptr(msd, ctypes.c_double))
    return msd
