"""Module-level configuration (mdsuite/utils/config.py:30-59 of the reference).

``memory_fraction`` is the only field the hot path reads: the memory manager
multiplies the available host memory by it to size the load batches, and the set
of correlation windows follows from that batch size.  The engine adds the device
choice; there is no CPU compute backend.
"""
from dataclasses import dataclass


@dataclass
class Config:
    jupyter: bool = False
    bokeh_sizing_mode: str = "stretch_both"
    memory_fraction: float = 0.5
    # engine additions
    device_index: int = 0
    strict_reference_masks: bool = True


config = Config()
