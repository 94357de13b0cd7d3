"""Integrated heat current ``sum_atoms r * (ke + pe)`` on the GPU
(mdsuite/transformations/integrated_heat_current.py:35-60 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import MultiSpeciesTrafo


class IntegratedHeatCurrent(MultiSpeciesTrafo):
    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.unwrapped_positions,
                              mdsuite_properties.kinetic_energy,
                              mdsuite_properties.potential_energy],
            output_property=mdsuite_properties.integrated_heat_current,
            scale_function={"linear": {"scale_factor": 2}},
        )

    def transform_batch(self, batch: typing.Dict[str, typing.Dict[str, np.ndarray]],
                        carryover: typing.Any = None) -> np.ndarray:
        out = None
        for properties in batch.values():
            arrays = [np.asarray(properties[p.name]) for p in self.input_properties]
            dtype = np.float32 if all(a.dtype == np.float32 for a in arrays) else np.float64
            arrays = [np.ascontiguousarray(a, dtype=dtype) for a in arrays]
            out = engine.energy_weighted_sum(*arrays, out=out, accumulate=out is not None)
        return out
