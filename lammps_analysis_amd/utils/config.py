"""Module-level configuration (mdsuite/utils/config.py:30-59 of the reference).

``memory_fraction`` is the only field the hot path reads: the memory manager
multiplies the available host memory by it to size the load batches, and the set
of correlation windows follows from that batch size.  The engine adds the device
choice: ``device_index`` is the GPU this process computes on, ``device_ids`` (optional)
the GPUs a multi-rank job may use -- rank r takes ``device_ids[r % len(device_ids)]``
instead of GPU ``local_rank`` (``parallel.init``).  ``backend`` names the compute
backend; ``"hip"`` is the only one: there is no CPU path to fall back to, and any other
value is rejected when a calculator selects its device.
"""
from dataclasses import dataclass
from typing import List, Optional


@dataclass
class Config:
    jupyter: bool = False
    bokeh_sizing_mode: str = "stretch_both"
    memory_fraction: float = 0.5
    # engine additions
    device_index: int = 0
    device_ids: Optional[List[int]] = None
    backend: str = "hip"
    strict_reference_masks: bool = True


config = Config()
