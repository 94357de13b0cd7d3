"""Multi-GPU sharding: one process per GPU, one exchange step per calculator run.

The units of work are independent: atom-pair tiles or frames for the RDF, time
origins (windows) for the correlations.  Every rank processes its share and the
results are combined once at the end: integer histograms and correlation sums by
one all-reduce, the per-window integral statistics by the parallel-variance merge.
``torch.distributed`` is used only as the transport (backend ``nccl`` is RCCL over
xGMI on ROCm; ``gloo`` on CPU for tests) and only if the caller has initialised
it; a single process needs neither torch nor a collective.

PyTorch-ROCm wheels carry their own HIP runtime.  A multi-GPU script must therefore
``import torch`` (and initialise the process group) before the first calculator or
engine call, which is when libmdx.so is loaded: the loader then binds libmdx to the
HIP runtime torch already brought in, instead of starting a second one.
"""
from __future__ import annotations

import sys
from typing import List, Tuple

import numpy as np


def _dist():
    torch = sys.modules.get("torch")
    if torch is None:
        return None
    dist = getattr(torch, "distributed", None)
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return None
    return dist


def world() -> Tuple[int, int]:
    """(rank, world size); (0, 1) when no process group is initialised."""
    dist = _dist()
    if dist is None:
        return 0, 1
    return dist.get_rank(), dist.get_world_size()


def split_range(n: int, rank: int, size: int) -> Tuple[int, int]:
    """Contiguous share ``[begin, end)`` of ``n`` units; sizes differ by at most one."""
    base, extra = divmod(int(n), int(size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def split_frames(frames: np.ndarray, rank: int, size: int) -> np.ndarray:
    """This rank's share of a list of frame indices."""
    begin, end = split_range(len(frames), rank, size)
    return frames[begin:end]


def _device_for_backend(dist):
    if dist.get_backend() == "nccl":
        import torch
        return torch.device("cuda", torch.cuda.current_device())
    return None


def all_reduce_sum(array: np.ndarray) -> np.ndarray:
    """Element-wise sum over ranks (identity without a process group).

    uint64 histograms travel as int64: counts stay far below 2**63.
    """
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return array
    import torch
    arr = np.ascontiguousarray(array)
    as_int = arr.dtype == np.uint64
    t = torch.from_numpy(arr.view(np.int64) if as_int else arr.copy())
    device = _device_for_backend(dist)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = t.cpu().numpy()
    return out.view(np.uint64) if as_int else out


def all_gather_objects(obj) -> List:
    """Every rank's object, in rank order."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def merge_moments(moments):
    """Combine every rank's window statistics (engine.Moments) in rank order."""
    merged = None
    for part in all_gather_objects(moments):
        if part is None:
            continue
        merged = part if merged is None else merged.merge(part)
    return merged
