"""Batch sizing from host memory.

Same arithmetic as mdsuite/memory_management/memory_manager.py:69-372 of the
reference.  It is part of the numerical contract of the correlation calculators:
windows never span two load batches, so the batch size decides which time origins
exist.  The constructor's ``memory_fraction`` argument is ignored in favour of
``config.memory_fraction``, as in the reference (:108).
"""
from __future__ import annotations

import logging
from typing import Tuple

import numpy as np

from ..utils.config import config
from ..utils.meta_functions import get_machine_properties
from ..utils.scale_functions import (linear_scale_function, linearithmic_scale_function,
                                     polynomial_scale_function, quadratic_scale_function)

log = logging.getLogger(__name__)

_SCALE_FUNCTIONS = {
    "linear": linear_scale_function,
    "log-linear": linearithmic_scale_function,
    "quadratic": quadratic_scale_function,
    "polynomial": polynomial_scale_function,
}

# atom-wise mini-batch fractions tried in order (:294)
_FRACTIONS = [1 / 2, 1 / 4, 1 / 8, 1 / 20, 1 / 100, 1 / 200, 0]


class MemoryManager:
    """Decides how many configurations (and, if needed, atoms) fit in one batch."""

    def __init__(self, data_path: list = None, database=None, parallel: bool = False,
                 memory_fraction: float = 0.2, scale_function: dict = None,
                 gpu: bool = False, offset: int = 0):
        if scale_function is None:
            scale_function = {"linear": {"scale_factor": 10}}
        self.data_path = data_path
        self.parallel = parallel
        self.database = database
        self.memory_fraction = config.memory_fraction
        self.offset = offset
        self.machine_properties = get_machine_properties()

        self.batch_size = None
        self.n_batches = None
        self.remainder = None
        self.atom_batch_size = None
        self.n_atom_batches = None
        self.atom_remainder = None
        self.minibatch = False
        self.scale_function, self.scale_function_parameters = \
            self._select_scale_function(scale_function)

    @staticmethod
    def _select_scale_function(input_dict: dict):
        name = list(input_dict.keys())[0]
        try:
            return _SCALE_FUNCTIONS[name], input_dict[name]
        except KeyError:
            raise KeyError("Invalid choice")

    def _budget(self) -> float:
        return self.memory_fraction * self.machine_properties["memory"]

    def get_batch_size(self) -> tuple:
        """(batch_size, number_of_batches, remainder) in configurations."""
        if self.data_path is None:
            raise ValueError("No tensor_values have been requested.")
        per_configuration_memory = 0.0
        n_configs = 0
        for item in self.data_path:
            _, n_configs, n_bytes = self.database.get_data_size(item)
            per_configuration_memory += n_bytes / n_configs
        per_configuration_memory = self.scale_function(per_configuration_memory,
                                                       **self.scale_function_parameters)
        maximum_loaded = int(np.clip(self._budget() / per_configuration_memory, 1,
                                     n_configs - self.offset))
        batch_size = self._get_optimal_batch_size(maximum_loaded)
        number_of_batches, remainder = divmod(n_configs - self.offset, batch_size)
        self.batch_size = batch_size
        self.n_batches = number_of_batches
        self.remainder = remainder
        return batch_size, number_of_batches, remainder

    @staticmethod
    def hdf5_load_time(n: int):
        return np.log(n)

    @staticmethod
    def _get_optimal_batch_size(naive_size):
        return naive_size

    def _compute_atomwise_minibatch(self, data_range: int):
        """Fallback when one window does not fit: batch over atoms as well."""
        per_atom_memory = 0
        per_configuration_memory = 0
        n_particles = n_configs = 0
        for item in self.data_path:
            n_particles, n_configs, n_bytes = self.database.get_data_size(item)
            per_configuration_memory += n_bytes / n_configs
            per_atom_memory += per_configuration_memory / n_particles
        per_atom_memory = self.scale_function(per_atom_memory, **self.scale_function_parameters)
        batch_size = None
        for fraction in _FRACTIONS:
            if fraction == 0:
                log.info("Could not find a good mini batch fraction - using single atom "
                         "mini batching!")
                batch_size = int(np.clip(self._budget() / per_atom_memory, 1, n_configs))
                self.atom_batch_size = 1
                break
            batch_size = int(np.clip(self._budget() / (fraction * per_atom_memory), 1,
                                     n_configs))
            if batch_size > data_range:
                self.atom_batch_size = n_particles * fraction
                break
        self.batch_size = batch_size
        self.n_batches = int(n_configs / batch_size)
        self.remainder = int(n_configs % batch_size)
        self.n_atom_batches = int(n_particles / self.atom_batch_size)
        self.atom_remainder = int(n_particles % self.atom_batch_size)

    def get_ensemble_loop(self, data_range: int, correlation_time: int = 1) -> Tuple[int, bool]:
        """(windows per batch, atom-wise mini-batching needed)."""
        final_window = self.batch_size - data_range
        if final_window < 0:
            self._compute_atomwise_minibatch(data_range)
            final_window = self.batch_size - data_range
            self.minibatch = True
            return int(np.clip(final_window / correlation_time, 1, None)), True
        return int(np.clip(final_window / correlation_time, 1, None)), False
