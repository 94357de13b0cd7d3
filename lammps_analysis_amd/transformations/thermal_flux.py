"""Thermal flux ``sum_atoms ((ke + pe) v - S.v)`` on the GPU
(mdsuite/transformations/thermal_flux.py:36-92 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import MultiSpeciesTrafo


class ThermalFlux(MultiSpeciesTrafo):
    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.stress, mdsuite_properties.velocities,
                              mdsuite_properties.kinetic_energy,
                              mdsuite_properties.potential_energy],
            output_property=mdsuite_properties.thermal_flux,
            scale_function={"linear": {"scale_factor": 5}},
        )

    def transform_batch(self, batch: typing.Dict[str, typing.Dict[str, np.ndarray]],
                        carryover: typing.Any = None) -> np.ndarray:
        out = None
        for properties in batch.values():
            arrays = [np.ascontiguousarray(properties[p.name], dtype=np.float32)
                      if np.asarray(properties[p.name]).dtype == np.float32
                      else np.ascontiguousarray(properties[p.name], dtype=np.float64)
                      for p in self.input_properties]
            if len({a.dtype for a in arrays}) != 1:
                arrays = [a.astype(np.float64) for a in arrays]
            out = engine.thermal_flux(*arrays, out=out, accumulate=out is not None)
        return out
