"""Wrap coordinates into the simulation box on the GPU
(mdsuite/transformations/wrap_coordinates.py:35-74 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import SingleSpeciesTrafo


class CoordinateWrapper(SingleSpeciesTrafo):
    def __init__(self, center_box: bool = True):
        """center_box True (default): wrap to [-L/2, L/2]; False: wrap to [0, L]."""
        super().__init__(
            input_properties=[mdsuite_properties.unwrapped_positions,
                              mdsuite_properties.box_length],
            output_property=mdsuite_properties.positions,
            scale_function={"linear": {"scale_factor": 2}},
        )
        self.center_box: bool = center_box

    def transform_batch(self, batch: typing.Dict[str, np.ndarray],
                        carryover: typing.Any = None) -> np.ndarray:
        unwrap_pos = batch[mdsuite_properties.unwrapped_positions.name]
        box_l = batch[mdsuite_properties.box_length.name]
        op = engine.MAP_WRAP_CENTERED if self.center_box else engine.MAP_WRAP
        return engine.map_coordinates(op, unwrap_pos, box_l)
