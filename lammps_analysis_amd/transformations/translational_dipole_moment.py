"""Translational dipole moment ``M(t) = sum q * r_unwrapped`` on the GPU
(mdsuite/transformations/translational_dipole_moment.py:35-62 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import MultiSpeciesTrafo, weight_arguments


class TranslationalDipoleMoment(MultiSpeciesTrafo):
    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.unwrapped_positions,
                              mdsuite_properties.charge],
            output_property=mdsuite_properties.translational_dipole_moment,
            scale_function={"linear": {"scale_factor": 2}},
        )

    def transform_batch(self, batch: typing.Dict[str, typing.Dict[str, np.ndarray]],
                        carryover: typing.Any = None) -> np.ndarray:
        out = None
        for properties in batch.values():
            pos = properties[mdsuite_properties.unwrapped_positions.name]
            scalar, weights = weight_arguments(properties[mdsuite_properties.charge.name],
                                               pos.shape[0])
            out = engine.weighted_atom_sum(pos, scalar_weight=scalar, weights=weights,
                                           out=out, accumulate=out is not None)
        return out
