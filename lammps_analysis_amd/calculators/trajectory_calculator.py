"""Base class of the calculators that stream the trajectory.

Same responsibilities as mdsuite/calculators/trajectory_calculator.py:49-406 of the
reference: lazy database handle, dependency resolution through transformations,
``tau_values`` handling, memory/data managers, and the batch iterator.  The
reference wraps its generators in ``tf.data`` with a prefetch thread; here a
background thread loads the next batch while the GPU works on the current one.
"""
from __future__ import annotations

import queue
import threading
from abc import ABC
from typing import List, Union

import numpy as np

from ..database.data_manager import DataManager
from ..database.simulation_database import (Database, PropertyInfo,  # noqa: F401
                                            open_database)
from ..memory_management.memory_manager import MemoryManager
from ..transformations import switcher_transformations
from ..utils.meta_functions import join_path
from .calculator import Calculator


class PrefetchIterator:
    """Runs a generator in a background thread, one item ahead of the consumer."""

    _END = object()

    def __init__(self, generator, depth: int = 2):
        self._queue = queue.Queue(maxsize=depth)
        self._error = None

        def worker():
            try:
                for item in generator:
                    self._queue.put(item)
            except BaseException as exc:  # re-raised in the consumer
                self._error = exc
            finally:
                self._queue.put(self._END)

        self._thread = threading.Thread(target=worker, daemon=True)
        self._thread.start()

    def __iter__(self):
        return self

    def __next__(self):
        item = self._queue.get()
        if item is self._END:
            if self._error is not None:
                raise self._error
            raise StopIteration
        return item


class TrajectoryCalculator(Calculator, ABC):
    def __init__(self, experiment=None, experiments: List = None):
        super().__init__(experiment=experiment, experiments=experiments)
        self.data_resolution = None
        self.loaded_property: PropertyInfo = None
        self.dependency: PropertyInfo = None
        self.scale_function = None
        self.batch_size: int = None
        self.n_batches: int = None
        self.remainder: int = None
        self.minibatch: bool = None
        self.ensemble_loop: int = None
        self.memory_manager = None
        self.data_manager = None
        self._database = None

    @property
    def database(self):
        if self._database is None:
            self._database = open_database(self.experiment.database_path / "database.hdf5")
        return self._database

    # ----------------------------------------------------------- dependencies

    def _run_dependency_check(self):
        if self.loaded_property is None:
            return
        if self.dependency is not None:
            if not self.database.check_existence(self.dependency.name):
                self._resolve_dependencies(self.dependency)
        if not self.database.check_existence(self.loaded_property.name):
            self._resolve_dependencies(self.loaded_property)

    def _resolve_dependencies(self, dependency: PropertyInfo):
        """Produce a missing property through its transformation, if one exists."""
        try:
            name = switcher_transformations[dependency.name]
        except KeyError:
            raise KeyError("Data not in database and cannot be generated.")
        getattr(self.experiment.run, name)()

    # -------------------------------------------------------------- tau / time

    def _handle_tau_values(self) -> np.ndarray:
        """Normalise ``args.tau_values`` (int, list/array or slice) to an index
        array, update ``args.data_range`` / ``data_resolution`` and return the times
        (trajectory_calculator.py:196-228 of the reference)."""
        if isinstance(self.args.tau_values, (int, np.integer)):
            self.data_resolution = self.args.tau_values
            self.args.tau_values = np.linspace(0, self.args.data_range - 1,
                                               self.args.tau_values, dtype=int)
        if isinstance(self.args.tau_values, (list, np.ndarray)):
            self.data_resolution = len(self.args.tau_values)
            self.args.data_range = self.args.tau_values[-1] + 1
        if isinstance(self.args.tau_values, slice):
            self.args.tau_values = np.linspace(0, self.args.data_range - 1,
                                               self.args.data_range,
                                               dtype=int)[self.args.tau_values]
            self.data_resolution = len(self.args.tau_values)
        return (np.asarray(self.args.tau_values) * self.experiment.time_step
                * self.experiment.sample_rate)

    # ---------------------------------------------------------------- managers

    def _check_remainder(self):
        return self.remainder - (self.remainder % self.args.data_range)

    def _prepare_managers(self, data_path: list, correct: bool = False):
        self.memory_manager = MemoryManager(data_path=data_path, database=self.database,
                                            memory_fraction=0.8,
                                            scale_function=self.scale_function)
        self.batch_size, self.n_batches, self.remainder = \
            self.memory_manager.get_batch_size()
        self.ensemble_loop, self.minibatch = self.memory_manager.get_ensemble_loop(
            self.args.data_range, self.args.correlation_time)
        if self.minibatch:
            self.batch_size = self.memory_manager.batch_size
            self.n_batches = self.memory_manager.n_batches
            self.remainder = self.memory_manager.remainder
        self._check_remainder()
        if correct:
            self._correct_batch_properties()
        self.data_manager = DataManager(
            data_path=data_path, database=self.database, data_range=self.args.data_range,
            batch_size=self.batch_size, n_batches=self.n_batches,
            ensemble_loop=self.ensemble_loop, correlation_time=self.args.correlation_time,
            remainder=self.remainder, atom_selection=self.args.atom_selection,
            minibatch=self.minibatch, atom_batch_size=self.memory_manager.atom_batch_size,
            n_atom_batches=self.memory_manager.n_atom_batches,
            atom_remainder=self.memory_manager.atom_remainder)

    def _correct_batch_properties(self):
        raise NotImplementedError

    def get_batch_dataset(self, subject_list: list = None, loop_array: np.ndarray = None,
                          correct: bool = False):
        """Iterator over load batches: dicts ``{b"<subject>/<Property>": array
        [n_atoms, n_frames, n_dims], b"data_size": int}`` (the hand-off format of
        trajectory_calculator.py:346-363 of the reference)."""
        path_list = [join_path(item, self.loaded_property.name) for item in subject_list]
        self._prepare_managers(path_list, correct=correct)
        generator, args = self.data_manager.batch_generator(system=self.system_property,
                                                            loop_array=loop_array)

        def encoded():
            for batch in generator(*args):
                yield {(k.encode() if isinstance(k, str) else k): v for k, v in batch.items()}

        return PrefetchIterator(encoded())

    def get_ensemble_dataset(self, batch: dict, subject: Union[str, list]):
        """Host-side iterator over the windows of a batch (reference semantics); the
        calculators of this engine pass whole batches to the GPU instead."""
        generator, args = self.data_manager.ensemble_generator(
            glob_data=batch, system=self.system_property)
        return generator(*args)

    def batch_windows(self, batch_array: np.ndarray, data_size: int) -> int:
        """Number of windows of a batch that the reference would evaluate and that the
        loaded array can supply in full."""
        n_win = self.data_manager.window_count(int(data_size))
        frames = batch_array.shape[1]
        if frames < self.args.data_range:
            return 0
        admissible = (frames - self.args.data_range) // self.args.correlation_time + 1
        return min(n_win, admissible)
