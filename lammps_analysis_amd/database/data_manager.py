"""Batch and window generators.

Follows mdsuite/database/data_manager.py:41-341 of the reference: a batch
generator that loads frame ranges (or explicit frame lists) of the requested
datasets, an atom-wise variant for mini-batching, and the description of the
sliding windows inside one batch.  The reference slices every window on the host
and hands it to TensorFlow; here the window loop itself runs on the GPU, so
``ensemble_generator`` additionally exposes ``window_count`` (the same count the
reference's ``dictionary_generator`` computes) for the kernels.
"""
from __future__ import annotations

import numpy as np

from .simulation_database import Database, open_database  # noqa: F401


class DataManager:
    def __init__(self, database: Database = None, data_path: list = None,
                 data_range: int = None, n_batches: int = None, batch_size: int = None,
                 ensemble_loop: int = None, correlation_time: int = 1,
                 remainder: int = None, atom_selection=np.s_[:], minibatch: bool = False,
                 atom_batch_size: int = None, n_atom_batches: int = None,
                 atom_remainder: int = None, offset: int = 0):
        self.database = database
        self.data_path = data_path
        self.minibatch = minibatch
        self.atom_batch_size = atom_batch_size
        self.n_atom_batches = n_atom_batches
        self.atom_remainder = atom_remainder
        self.offset = offset
        self.data_range = data_range
        self.n_batches = n_batches
        self.batch_size = batch_size
        self.remainder = remainder
        self.ensemble_loop = ensemble_loop
        self.correlation_time = correlation_time
        self.atom_selection = atom_selection

    # ------------------------------------------------------------------ batches

    def _frame_range(self, batch: int, batch_number: int, batch_size: int):
        start = int(batch * batch_size) + self.offset
        stop = int(start + batch_size)
        data_size = int(batch_size)
        if batch == batch_number:  # the remainder batch
            stop = int(start + self.remainder)
            data_size = int(self.remainder)
        return start, stop, data_size

    def batch_generator(self, dictionary: bool = False, system: bool = False,
                        remainder: bool = False, loop_array: np.ndarray = None) -> tuple:
        """Return ``(generator_function, args)`` like the reference.

        ``system=True`` slices axis 0 (data_manager.py:204-205): system series are
        stored ``(1, T, k)``, so batch 0 receives the whole series and later batches
        nothing; they always fit one batch in practice.
        """
        args = (self.n_batches, self.batch_size, self.database.path, self.data_path,
                dictionary)

        def generator(batch_number, batch_size, database, data_path, dictionary):
            database = open_database(database)
            loop_over_remainder = self.remainder is not None and self.remainder > 0
            for batch in range(batch_number + int(loop_over_remainder)):
                start, stop, data_size = self._frame_range(batch, batch_number, batch_size)
                if loop_array is not None:
                    if isinstance(self.atom_selection, dict):
                        select_slice = {item: np.s_[sel, loop_array[batch]]
                                        for item, sel in self.atom_selection.items()}
                    else:
                        select_slice = np.s_[self.atom_selection, loop_array[batch]]
                elif system:
                    if start > 0:
                        # the axis-0 slice of a (1, T, k) series is empty from the second
                        # batch on (the reference would hand an empty tensor to the
                        # calculator and fail there); batch 0 carried the whole series
                        continue
                    select_slice = np.s_[start:stop]
                elif isinstance(self.atom_selection, dict):
                    select_slice = {item: np.s_[sel, start:stop]
                                    for item, sel in self.atom_selection.items()}
                else:
                    select_slice = np.s_[self.atom_selection, start:stop]
                yield database.load_data(data_path, select_slice=select_slice,
                                         dictionary=dictionary, d_size=data_size)

        def atom_generator(batch_number, batch_size, database, data_path, dictionary):
            if isinstance(self.atom_selection, dict):
                raise ValueError("Atom selection is not currently available "
                                 "for mini-batched calculations")
            database = open_database(database)
            extra = 1 if self.atom_remainder else 0
            start = 0
            for atom_batch in range(self.n_atom_batches + extra):
                atom_start = atom_batch * self.atom_batch_size
                atom_stop = atom_start + self.atom_batch_size
                if atom_batch == self.n_atom_batches:
                    atom_stop = start + self.atom_remainder
                for batch in range(batch_number + int(remainder)):
                    start, stop, data_size = self._frame_range(batch, batch_number,
                                                               batch_size)
                    select_slice = np.s_[int(atom_start):int(atom_stop), start:stop]
                    yield database.load_data(data_path, select_slice=select_slice,
                                             dictionary=dictionary, d_size=data_size)

        return (atom_generator if self.minibatch else generator), args

    # ------------------------------------------------------------------ windows

    def window_count(self, data_size: int) -> int:
        """Windows of one batch: ``int(clip((data_size - data_range) / ct, 1, None))``
        (data_manager.py:324-328).  The last admissible origin is excluded."""
        return int(np.clip((data_size - self.data_range) / self.correlation_time, 1, None))

    def ensemble_generator(self, system: bool = False, glob_data: dict = None) -> tuple:
        """Host-side window slicer with the reference's semantics, kept for callers
        that want the windows themselves (tests, custom calculators)."""
        args = (self.ensemble_loop, self.correlation_time, self.data_range)

        def dictionary_generator(ensemble_loop, correlation_time, data_range):
            ensemble_loop = int(np.clip(
                (glob_data[b"data_size"] - data_range) / correlation_time, 1, None))
            for ensemble in range(ensemble_loop):
                start = ensemble * correlation_time
                stop = start + data_range
                yield {key: value[:, start:stop] for key, value in glob_data.items()
                       if key != b"data_size"}

        return dictionary_generator, args
