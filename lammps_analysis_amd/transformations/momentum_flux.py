"""Momentum flux: per-frame sum of the off-diagonal stress components on the GPU
(mdsuite/transformations/momentum_flux.py:35-55 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import MultiSpeciesTrafo


class MomentumFlux(MultiSpeciesTrafo):
    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.stress],
            output_property=mdsuite_properties.momentum_flux,
            scale_function={"linear": {"scale_factor": 4}},
        )

    def transform_batch(self, batch: typing.Dict[str, typing.Dict[str, np.ndarray]],
                        carryover: typing.Any = None) -> np.ndarray:
        out = None
        for properties in batch.values():
            stress = properties[mdsuite_properties.stress.name]
            out = engine.momentum_flux(stress, out=out, accumulate=out is not None)
        return out
