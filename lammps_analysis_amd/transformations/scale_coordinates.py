"""Scale fractional coordinates by the box lengths on the GPU
(mdsuite/transformations/scale_coordinates.py:35-54 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import SingleSpeciesTrafo


class ScaleCoordinates(SingleSpeciesTrafo):
    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.scaled_positions,
                              mdsuite_properties.box_length],
            output_property=mdsuite_properties.positions,
            scale_function={"linear": {"scale_factor": 2}},
        )

    def transform_batch(self, batch: typing.Dict[str, np.ndarray],
                        carryover: typing.Any = None) -> np.ndarray:
        pos = batch[mdsuite_properties.scaled_positions.name]
        box_l = batch[mdsuite_properties.box_length.name]
        return engine.map_coordinates(engine.MAP_SCALE, pos, box_l)
