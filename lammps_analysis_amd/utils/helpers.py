"""Small helpers of the host code: the "argument not given" sentinel and the memory
budget switch that the reference's tests use (mdsuite/utils/helpers.py:20-83)."""
from __future__ import annotations

from .meta_functions import get_machine_properties

_GIB = float(1 << 30)
_LARGEST_FRACTION = 0.9


class NoneType:
    """Stands for "argument not given" where ``None`` itself means something.  The class
    is the marker; it is never instantiated."""

    def __init__(self):
        raise NotImplementedError()


def compute_memory_fraction(desired_memory: float, total_memory: float = None):
    """``desired_memory`` (GB) as a fraction of the host memory (``total_memory`` GB, by
    default what the machine reports), at most 0.9."""
    available = total_memory if total_memory is not None \
        else get_machine_properties()["memory"] / _GIB
    return min(desired_memory / available, _LARGEST_FRACTION)


class change_memory_fraction:
    """``with change_memory_fraction(gb):`` -- run the block with ``config.memory_fraction``
    set for a budget of ``gb`` GB (``None``: leave it alone); the previous value comes back
    on the way out, also after an exception."""

    def __init__(self, desired_memory=None):
        self._wanted = desired_memory
        self._kept = None

    def __enter__(self):
        from .config import config

        self._kept = config.memory_fraction
        if self._wanted is not None:
            config.memory_fraction = compute_memory_fraction(self._wanted)
        return self

    def __exit__(self, *exc_info):
        from .config import config

        config.memory_fraction = self._kept
        return False
