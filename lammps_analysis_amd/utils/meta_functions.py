"""Small host helpers (subset of mdsuite/utils/meta_functions.py of the reference
that the hot path uses: :54-93, :96-129, :132-158)."""
from __future__ import annotations

import json
import os

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
