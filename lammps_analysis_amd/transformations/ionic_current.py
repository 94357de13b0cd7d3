"""Ionic current ``J(t) = sum_species sum_atoms q * v`` on the GPU
(mdsuite/transformations/ionic_current.py:35-58 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import MultiSpeciesTrafo, weight_arguments


class IonicCurrent(MultiSpeciesTrafo):
    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.velocities, mdsuite_properties.charge],
            output_property=mdsuite_properties.ionic_current,
            scale_function={"linear": {"scale_factor": 2}},
        )

    def transform_batch(self, batch: typing.Dict[str, typing.Dict[str, np.ndarray]],
                        carryover: typing.Any = None) -> np.ndarray:
        out = None
        for properties in batch.values():
            vel = properties[mdsuite_properties.velocities.name]
            scalar, weights = weight_arguments(properties[mdsuite_properties.charge.name],
                                               vel.shape[0])
            out = engine.weighted_atom_sum(vel, scalar_weight=scalar, weights=weights,
                                           out=out, accumulate=out is not None)
        return out
