"""Unwrap coordinates with stored image indices, ``pos + box_image * box_length``, on
the GPU (mdsuite/transformations/unwrap_via_indices.py:35-57 of the reference)."""
import typing

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from .transformations import SingleSpeciesTrafo


class UnwrapViaIndices(SingleSpeciesTrafo):
    def __init__(self):
        super().__init__(
            input_properties=[mdsuite_properties.positions, mdsuite_properties.box_length,
                              mdsuite_properties.box_images],
            output_property=mdsuite_properties.unwrapped_positions,
            scale_function={"linear": {"scale_factor": 2}},
        )

    def transform_batch(self, batch: typing.Dict[str, np.ndarray],
                        carryover: typing.Any = None) -> np.ndarray:
        pos = batch[mdsuite_properties.positions.name]
        box_im = batch[mdsuite_properties.box_images.name]
        box_l = batch[mdsuite_properties.box_length.name]
        return engine.map_coordinates(engine.MAP_UNWRAP_INDICES, pos, box_l, aux=box_im)
