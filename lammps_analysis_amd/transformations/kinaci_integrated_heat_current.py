"""Kinaci integrated heat current on the GPU
(mdsuite/transformations/kinaci_integrated_heat_current.py:35-87 of the reference):
``sum_atoms r * (integral of F.v dt + pe)``, the running integral carried from batch to
batch.  The integral is kept per atom of each species; the reference text adds the
integral arrays of all species element by element, which only runs for species of equal
size and has no test, and is not reproduced.  Parity unpinned."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import MultiSpeciesTrafo


class KinaciIntegratedHeatCurrent(MultiSpeciesTrafo):
    # a frame's integral needs every earlier frame: batches run whole, on every rank
    frame_shardable = False

    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.unwrapped_positions,
                              mdsuite_properties.velocities,
                              mdsuite_properties.forces,
                              mdsuite_properties.potential_energy,
                              mdsuite_properties.time_step,
                              mdsuite_properties.sample_rate],
            output_property=mdsuite_properties.kinaci_heat_current,
            scale_function={"linear": {"scale_factor": 5}},
        )

    def transform_batch(self, batch: typing.Dict[str, typing.Dict[str, np.ndarray]],
                        carryover: typing.Any = None
                        ) -> typing.Tuple[np.ndarray, typing.Dict[str, np.ndarray]]:
        out = None
        carry_next = {}
        per_atom = self.input_properties[:4]
        for sp_name, properties in batch.items():
            arrays = [np.asarray(properties[p.name]) for p in per_atom]
            dtype = np.float32 if all(a.dtype == np.float32 for a in arrays) else np.float64
            arrays = [np.ascontiguousarray(a, dtype=dtype) for a in arrays]
            time_step = float(np.ravel(properties[mdsuite_properties.time_step.name])[0])
            sample_rate = float(np.ravel(properties[mdsuite_properties.sample_rate.name])[0])
            carry = None if carryover is None else carryover[sp_name]
            out, carry_next[sp_name] = engine.kinaci_heat_current(
                *arrays, dt=time_step * sample_rate, carry=carry, out=out,
                accumulate=out is not None)
        return out, carry_next
