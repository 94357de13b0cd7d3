"""Angular distribution function.

Host logic (arguments, frame sampling, batching, per-batch density normalisation,
result layout) follows mdsuite/calculators/angular_distribution_function.py:58-609 of
the reference; the per-batch body -- pair vectors, float16 cutoff test, triplet
enumeration, angles, weights, histogram -- is one call into the GPU engine
(``engine.adf_accumulate``).

Kept from the reference: every batch is normalised to a density on its own
(``np.histogram(..., density=True)``) and the float32 results are summed; only species
triples (i, j, k) in non-decreasing species order are counted; angles are labelled with
``180 / 3.14159``; ``start`` defaults to 1.
"""
from __future__ import annotations

import itertools
import logging
from abc import ABC
from dataclasses import dataclass

import numpy as np

from .. import engine
from ..database.mdsuite_properties import mdsuite_properties
from ..utils.config import config
from ..utils.meta_functions import join_path
from .calculator import call
from .trajectory_calculator import TrajectoryCalculator

log = logging.getLogger(__name__)


@dataclass
class Args:
    number_of_bins: int
    number_of_configurations: int
    correlation_time: int
    atom_selection: np.s_
    data_range: int
    cutoff: float
    start: int
    stop: int
    species: list
    molecules: bool
    norm_power: int


class AngularDistributionFunction(TrajectoryCalculator, ABC):
    """``experiment.run.AngularDistributionFunction(number_of_configurations=100,
    cutoff=3.2, number_of_bins=100)``"""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.scale_function = {"quadratic": {"outer_scale_factor": 10}}
        self.loaded_property = mdsuite_properties.positions
        self.use_tf_function = None
        self.molecules = None
        self.bin_range = None
        self.number_of_atoms = None
        self.norm_power = None
        self.sample_configurations = None
        self.result_keys = ["max_peak"]
        self.result_series_keys = ["angle", "adf"]
        self._dtype = np.float32
        self.adf_minibatch = None
        self.override_n_batches = None
        self.analysis_name = "Angular_Distribution_Function"
        self.x_label = r"$$\text{Angle} / \theta$$"
        self.y_label = r"$$\text{ADF} / a.u.$$"

    @call
    def __call__(self, batch_size: int = 1, minibatch: int = -1,
                 number_of_configurations: int = 5, cutoff: float = 6.0, start: int = 1,
                 stop: int = None, number_of_bins: int = 500, species: list = None,
                 use_tf_function: bool = False, molecules: bool = False,
                 atom_selection=np.s_[:], plot: bool = True, norm_power: int = 4, **kwargs):
        """Same arguments as the reference (:135-213); ``minibatch`` and
        ``use_tf_function`` steer the reference's memory use and have no effect here."""
        self.args = Args(number_of_bins=number_of_bins, cutoff=cutoff, start=start, stop=stop,
                         atom_selection=atom_selection, data_range=1, correlation_time=1,
                         molecules=molecules, species=species,
                         number_of_configurations=number_of_configurations,
                         norm_power=norm_power)
        self.use_tf_function = use_tf_function
        self.cutoff = cutoff
        self.plot = plot
        self._batch_size = batch_size
        self.adf_minibatch = minibatch
        self.bin_range = [0.0, 3.15]  # from 0 to a chemist's pi
        self.norm_power = norm_power
        self.override_n_batches = kwargs.get("batches")

    def check_input(self):
        self._run_dependency_check()
        if self.args.stop is None:
            self.args.stop = self.experiment.number_of_configurations - 1
        if self.args.species is None:
            if self.args.molecules:
                self.args.species = list(self.experiment.molecules)
            else:
                self.args.species = list(self.experiment.species)

    def _prepare_data_structure(self):
        return np.linspace(self.args.start, self.args.stop,
                           self.args.number_of_configurations, dtype=int)

    def _correct_batch_properties(self):
        """Batching of a structural calculator (:520-548 of the reference)."""
        if self.batch_size > self.args.number_of_configurations:
            self.batch_size = self.args.number_of_configurations
            self.n_batches = 1
        else:
            self.n_batches = int(self.args.number_of_configurations / self.batch_size)
        if self.override_n_batches is not None:
            self.n_batches = self.override_n_batches
        if self.minibatch:
            self.batch_size = 1
            self.n_batches = self.args.number_of_configurations
            self.memory_manager.atom_batch_size = None
            self.memory_manager.n_atom_batches = None
            self.memory_manager.atom_remainder = None
        self.remainder = 0
        self.minibatch = False

    def prepare_computation(self):
        path_list = [join_path(name, self.loaded_property.name) for name in self.args.species]
        self._prepare_managers(path_list)
        self._correct_batch_properties()
        dict_keys = [str.encode(join_path(name, self.loaded_property.name))
                     for name in self.args.species]
        split_arr = np.array_split(self.sample_configurations, self.n_batches)
        return dict_keys, split_arr

    def _compute_adfs(self, angles: dict):
        index = list(range(len(self.args.species)))
        for combo in itertools.combinations_with_replacement(index, 3):
            names = [self.args.species[c] for c in combo]
            hist = angles["-".join(names)]
            bin_range_to_angles = np.linspace(self.bin_range[0] * (180 / 3.14159),
                                              self.bin_range[1] * (180 / 3.14159),
                                              self.args.number_of_bins)
            self.selected_species = names
            self.data_range = self.args.number_of_configurations
            max_angle = bin_range_to_angles[int(np.argmax(hist))]
            data = {
                self.result_keys[0]: float(max_angle),
                self.result_series_keys[0]: bin_range_to_angles.tolist(),
                self.result_series_keys[1]: hist.tolist(),
            }
            self.queue_data(data=data, subjects=self.selected_species)

    def run_calculator(self):
        self.check_input()
        self.sample_configurations = self._prepare_data_structure()
        dict_keys, split_arr = self.prepare_computation()
        engine.set_device(config.device_index)
        box = np.asarray(self.experiment.box_array, dtype=np.float32)
        # the bin edges np.histogram uses for float32 samples and this range
        edges = np.histogram_bin_edges(np.zeros(1, dtype=np.float32),
                                       bins=self.args.number_of_bins, range=self.bin_range)
        widths = np.diff(edges)
        index = list(range(len(self.args.species)))
        combos = list(itertools.combinations_with_replacement(index, 3))
        angles = {}
        batch_ds = self.get_batch_dataset(subject_list=self.args.species,
                                          loop_array=split_arr, correct=True)
        for batch in batch_ds:
            positions = [np.ascontiguousarray(batch[key], dtype=np.float32)
                         for key in dict_keys]
            weights, _ = engine.adf_accumulate(positions, box, self.cutoff, edges,
                                               norm_power=self.norm_power)
            for c, combo in enumerate(combos):
                name = "-".join(self.args.species[s] for s in combo)
                with np.errstate(divide="ignore", invalid="ignore"):
                    density = (weights[c] / widths / weights[c].sum()).astype(np.float32)
                angles[name] = angles[name] + density if name in angles else density
        self._compute_adfs(angles)
