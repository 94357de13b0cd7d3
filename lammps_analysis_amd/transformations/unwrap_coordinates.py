"""Unwrap coordinates by detecting jumps across the box between consecutive frames,
on the GPU (mdsuite/transformations/unwrap_coordinates.py:35-81 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import SingleSpeciesTrafo


class CoordinateUnwrapper(SingleSpeciesTrafo):
    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.positions, mdsuite_properties.box_length],
            output_property=mdsuite_properties.unwrapped_positions,
            scale_function={"linear": {"scale_factor": 2}},
        )

    def transform_batch(self, batch: typing.Dict[str, np.ndarray],
                        carryover: typing.Any = None) -> typing.Tuple[np.ndarray, dict]:
        pos = batch[mdsuite_properties.positions.name]
        box_l = batch[mdsuite_properties.box_length.name]
        return engine.unwrap_coordinates(pos, box_l, carry=carryover)
