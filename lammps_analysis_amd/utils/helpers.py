"""Helpers (mdsuite/utils/helpers.py:20-83 of the reference)."""
from __future__ import annotations

import contextlib

import numpy as np

from .meta_functions import get_machine_properties


class NoneType:
    """Sentinel for "argument not given" where ``None`` is a meaningful value."""

    def __init__(self):
        raise NotImplementedError()


def compute_memory_fraction(desired_memory: float, total_memory: float = None):
    """Fraction of the available host memory that ``desired_memory`` GB is, clipped
    to 0.9."""
    if total_memory is None:
        total_memory = get_machine_properties()["memory"] / (1024.0**3)
    return np.clip(desired_memory / total_memory, None, 0.9)


@contextlib.contextmanager
def change_memory_fraction(desired_memory):
    """Temporarily set ``config.memory_fraction`` from an amount in GB."""
    from .config import config

    default = config.memory_fraction
    if desired_memory is not None:
        config.memory_fraction = compute_memory_fraction(desired_memory)
    try:
        yield
    finally:
        config.memory_fraction = default
