"""Velocities by forward differences of unwrapped positions, ``v(t) = (x(t+dt) -
x(t)) / dt``, the last frame of a batch repeating the one before, on the GPU
(mdsuite/transformations/velocity_from_positions.py:35-58 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import SingleSpeciesTrafo


class VelocityFromPositions(SingleSpeciesTrafo):
    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.unwrapped_positions,
                              mdsuite_properties.time_step, mdsuite_properties.sample_rate],
            output_property=mdsuite_properties.velocities_from_positions,
            scale_function={"linear": {"scale_factor": 2}},
        )

    def transform_batch(self, batch: typing.Dict[str, np.ndarray],
                        carryover: typing.Any = None) -> np.ndarray:
        pos = batch[mdsuite_properties.unwrapped_positions.name]
        dt = (np.asarray(batch[mdsuite_properties.time_step.name], dtype=np.float64)
              * np.asarray(batch[mdsuite_properties.sample_rate.name], dtype=np.float64))
        return engine.map_coordinates(engine.MAP_VELOCITY, pos, [float(dt.reshape(-1)[0]), 0, 0])
