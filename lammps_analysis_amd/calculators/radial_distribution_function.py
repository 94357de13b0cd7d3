"""Radial distribution function.

Host logic (argument defaults, frame sampling, batching, prefactor, ideal-gas
correction, result layout) follows
mdsuite/calculators/radial_distribution_function.py:58-887 of the reference; the
per-batch body -- index lists, gathers, minimum image, masks, histogram -- is one
call into the GPU engine (``engine.RdfEngine.accumulate``), whose integer counts
are bit-identical to that sequence of float32 operations.
"""
from __future__ import annotations

import itertools
import logging
from abc import ABC
from dataclasses import dataclass
from typing import Union

import numpy as np

from .. import engine, parallel
from ..database.mdsuite_properties import mdsuite_properties
from ..utils.config import config
from ..utils.meta_functions import join_path
from .calculator import call
from .trajectory_calculator import TrajectoryCalculator

log = logging.getLogger(__name__)

# Device-side batching: frames are sent to the GPU in blocks of about this many
# bytes of float32 positions.  Integer counts do not depend on the blocking.
_DEVICE_BATCH_BYTES = 1 << 30
# stored arrays up to this size may go to the device whole (a dense, strided frame sample)
_DEVICE_RESIDENT_BYTES = 64 << 30


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


class RadialDistributionFunction(TrajectoryCalculator, ABC):
    """``project.run.RadialDistributionFunction(number_of_configurations=500)``"""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.scale_function = {"quadratic": {"outer_scale_factor": 10,
                                             "inner_scale_factor": 5}}
        self.loaded_property = mdsuite_properties.positions
        self.x_label = r"$$r / nm$$"
        self.y_label = r"$$g(r)$$"
        self.analysis_name = "Radial_Distribution_Function"
        self.result_series_keys = ["x", "y"]
        self._dtype = np.float32
        self.rdf_minibatch = None
        self.use_tf_function = None
        self.override_n_batches = None
        self.index_list = None
        self.sample_configurations = None
        self.key_list = None
        self.rdf = None
        self.strict_reference_masks = None
        self.histograms = None

    @call
    def __call__(self, plot: bool = True, number_of_bins: int = None, cutoff: float = None,
                 save: bool = True, start: int = 0, stop: int = None,
                 number_of_configurations: int = 500,
                 atom_selection: Union[np.s_, dict] = np.s_[:], minibatch: int = -1,
                 species: list = None, molecules: bool = False, **kwargs):
        """Same arguments as the reference (:139-213).  Engine-only keyword:
        ``strict_reference_masks`` (default ``config.strict_reference_masks`` = True)
        reproduces the reference's strict species masks, which leave the first atom of
        every species out of all pairs (:637-638); ``False`` counts every atom.
        ``shard`` says how several ranks (one per GPU) split the work: ``"tiles"`` -- every
        rank loads all sampled frames and evaluates its share of the atom-pair tiles;
        ``"frames"`` -- every rank loads and evaluates its share of the frames; ``"auto"``
        (default) -- tiles when there are at least eight tile pairs per rank, else
        frames.  The counts are the same either way."""
        self.args = Args(number_of_bins=number_of_bins, cutoff=cutoff, start=start,
                         stop=stop, atom_selection=atom_selection, data_range=1,
                         correlation_time=1, molecules=molecules, species=species,
                         number_of_configurations=number_of_configurations)
        self.rdf_minibatch = minibatch
        self.plot = plot
        self.use_tf_function = kwargs.pop("use_tf_function", False)
        self.override_n_batches = kwargs.get("batches")
        self.tqdm_limit = kwargs.pop("tqdm", 10)
        self.strict_reference_masks = kwargs.pop("strict_reference_masks",
                                                 config.strict_reference_masks)
        self.shard = kwargs.pop("shard", "auto")
        if self.shard not in ("auto", "tiles", "frames"):
            raise ValueError("shard must be 'auto', 'tiles' or 'frames'")

    # ------------------------------------------------------------------ set-up

    def check_input(self):
        if self.args.stop is None:
            self.args.stop = self.experiment.number_of_configurations - 1
        if self.args.cutoff is None:
            self.args.cutoff = self.experiment.box_array[0] / 2 - 0.1
        if self.args.number_of_configurations == -1:
            self.args.number_of_configurations = \
                self.experiment.number_of_configurations - 1
        if self.rdf_minibatch == -1:
            self.rdf_minibatch = self.args.number_of_configurations
        if self.args.number_of_bins is None:
            self.args.number_of_bins = int(self.args.cutoff / 0.01)
        if self.args.species is None:
            if self.args.molecules:
                self.args.species = list(self.experiment.molecules)
            else:
                self.args.species = list(self.experiment.species)
        self._initialize_rdf_parameters()

    def _initialize_rdf_parameters(self):
        self.bin_range = [0, self.args.cutoff]
        self.index_list = list(range(len(self.args.species)))
        self.sample_configurations = np.linspace(
            self.args.start, self.args.stop, self.args.number_of_configurations, dtype=int)
        self.key_list = [self._get_species_names(x) for x in
                         itertools.combinations_with_replacement(self.index_list, r=2)]
        self.rdf = {name: np.zeros(self.args.number_of_bins) for name in self.key_list}

    def _get_species_names(self, species_tuple: tuple) -> str:
        return f"{self.args.species[species_tuple[0]]}_{self.args.species[species_tuple[1]]}"

    @property
    def particles_list(self):
        if self.args.molecules:
            return [self.experiment.molecules[item].n_particles
                    for item in self.experiment.molecules]
        if isinstance(self.args.atom_selection, dict):
            return [len(self.args.atom_selection[item]) for item in self.args.species]
        return [self.experiment.species[item].n_particles for item in self.args.species]

    def _correct_batch_properties(self):
        """Batching of a structural (non-dynamic) calculator (:395-420)."""
        if self.batch_size > self.args.number_of_configurations:
            self.batch_size = self.args.number_of_configurations
            self.n_batches = 1
        else:
            self.n_batches = int(self.args.number_of_configurations / self.batch_size)
        if self.override_n_batches is not None:
            self.n_batches = self.override_n_batches
        elif self.experiment.box_array is not None:
            # The quadratic scale function above sizes batches for the reference's
            # materialised pair tensors (one frame at 10k atoms).  The GPU path holds
            # positions only, so frames are grouped into much larger load batches.
            bytes_per_frame = 12 * max(1, sum(self.particles_list))
            frames = max(self.batch_size, _DEVICE_BATCH_BYTES // bytes_per_frame)
            self.n_batches = max(1, -(-self.args.number_of_configurations // frames))
        if self.minibatch:
            self.batch_size = 1
            self.n_batches = self.args.number_of_configurations
            self.memory_manager.atom_batch_size = None
            self.memory_manager.n_atom_batches = None
            self.memory_manager.atom_remainder = None
            self.minibatch = False
        self.n_batches = max(1, min(self.n_batches, self.args.number_of_configurations))
        self.remainder = 0

    def prepare_computation(self):
        path_list = [join_path(item, self.loaded_property.name) for item in self.args.species]
        self._prepare_managers(path_list)
        self._correct_batch_properties()
        dict_keys = [str.encode(join_path(item, self.loaded_property.name))
                     for item in self.args.species]
        split_arr = np.array_split(self.sample_configurations, self.n_batches)
        batch_tqdm = self.tqdm_limit > self.n_batches
        return dict_keys, split_arr, batch_tqdm

    # --------------------------------------------------------------- prefactor

    def _calculate_prefactor(self, species: Union[str, tuple] = None):
        first, second = species.split("_")
        species_scale_factor = 2 if first == second else 1
        if self.args.molecules:
            n_first = self.experiment.molecules[first].n_particles
            n_second = self.experiment.molecules[second].n_particles
        elif isinstance(self.args.atom_selection, dict):
            n_first = len(self.args.atom_selection[first])
            n_second = len(self.args.atom_selection[second])
        else:
            n_first = self.experiment.species[first].n_particles
            n_second = self.experiment.species[second].n_particles
        rho = n_second / self.experiment.volume
        denominator = (self.args.number_of_configurations * rho * self.ideal_correction
                       * n_first)
        return species_scale_factor / denominator

    @property
    def ideal_correction(self) -> np.ndarray:
        """Ideal-gas shell volume per bin, with the finite-box corrections beyond half
        the box length, evaluated on the un-normalised radii as the reference does
        (:719-826)."""
        box = self.experiment.box_array[0]
        radii = np.linspace(0.0, self.args.cutoff, self.args.number_of_bins)
        bin_width = self.args.cutoff / self.args.number_of_bins
        lower_bound = box / 2
        middle_bound = np.sqrt(2) * box / 2

        inner = radii[radii <= lower_bound]
        outer = radii[~(radii <= lower_bound)]
        parts = [4 * np.pi * inner**2]
        if outer.size:
            mid = outer[outer < middle_bound]
            far = outer[~(outer < middle_bound)]
            parts.append(2 * np.pi * mid * (3 - 4 * mid))
            if far.size:
                root = np.sqrt(4 * far**2 - 2)
                term = 8 * far * np.arctan((2 * far * (4 * far**2 - 3))
                                           / (root * (4 * far**2 + 1)))
                parts.append(2 * far * (3 * np.pi - 12 * np.arctan(root) + term))
        return np.concatenate(parts) * bin_width

    def _ang_to_nm(self, data_in: np.ndarray) -> np.ndarray:
        return (self.experiment.units.length / 1e-9) * data_in

    def _calculate_radial_distribution_functions(self):
        for names in self.key_list:
            self.selected_species = names.split("_")
            with np.errstate(divide="ignore", invalid="ignore"):
                prefactor = self._calculate_prefactor(names)
                self.rdf[names] = np.array(self.rdf[names], dtype=float) * prefactor
            x_data = self._ang_to_nm(
                np.linspace(0.0, self.args.cutoff, self.args.number_of_bins))
            data = {self.result_series_keys[0]: x_data.tolist(),
                    self.result_series_keys[1]: self.rdf[names].tolist()}
            self.queue_data(data=data, subjects=self.selected_species)

    # --------------------------------------------------------------------- run

    def _accumulate_from_the_store(self, rdf_engine, rank: int, size: int) -> bool:
        """Hand the stored arrays themselves to the engine when that is cheaper than
        gathering the sampled frames on the host first.

        Whole species and a sample of at least one frame in 32 only: long runs of consecutive frames are
        streamed from the memory maps block by block (no host copy at all); a strided but
        dense sample is sent with its frame list, the whole array going to the device.
        Histogram counts do not depend on how frames are grouped, so the result equals the
        batch loop's.  Returns False when the batch loop should run instead."""
        if self.args.molecules or not (isinstance(self.args.atom_selection, slice)
                                       and self.args.atom_selection == slice(None)):
            return False
        sample = np.asarray(self.sample_configurations, dtype=np.int64)
        if sample.size == 0 or np.any(np.diff(sample) < 0):
            return False
        paths = [join_path(item, self.loaded_property.name) for item in self.args.species]
        data = self.database.load_data(paths, select_slice=np.s_[:])
        views = [data[p] for p in paths]
        if any(v.ndim != 3 or v.dtype != np.float32 or not v.flags.c_contiguous for v in views):
            return False
        n_stored = views[0].shape[1]
        if sample[-1] >= n_stored:
            return False
        # maximal runs of consecutive frame indices
        breaks = np.flatnonzero(np.diff(sample) != 1) + 1
        runs = np.split(sample, breaks)
        # A few runs that are long on the whole: ``number_of_configurations=-1`` is one run
        # of n-2 frames and the last frame alone, because the truncating linspace leaves
        # out the frame before it.
        in_long_runs = sum(len(r) for r in runs if len(r) >= 128)
        if len(runs) <= 16 and 20 * in_long_runs >= 19 * sample.size:
            for run in runs:
                rdf_engine.accumulate(views, frame_range=(int(run[0]), int(run[-1]) + 1),
                                      shard_rank=rank, shard_count=size)
            return True
        total_bytes = sum(v.nbytes for v in views)
        # the upload of a whole array (53 GB/s) beats a host-side gather of the sampled frames
        # (about 1.5 GB/s through a fancy index) down to one frame in 32
        # ... as long as the arrays fit beside the packed frames: half of the device memory
        budget = min(_DEVICE_RESIDENT_BYTES, engine.device_info()["memory_bytes"] // 2)
        if 32 * sample.size >= n_stored and total_bytes <= budget:
            try:
                rdf_engine.accumulate(views, frames=sample, shard_rank=rank, shard_count=size)
            except MemoryError:
                # another process holds the memory: start over with the batch loop
                log.warning("device memory for the stored arrays is not available; "
                            "loading frame batches instead")
                engine.release_cached_memory()
                rdf_engine.reset()
                return False
            return True
        return False

    def run_calculator(self):
        self.check_input()
        dict_keys, split_arr, _ = self.prepare_computation()
        if isinstance(self.args.atom_selection, dict):
            particles = [len(self.args.atom_selection[item]) for item in self.args.species]
        else:
            particles = self.particles_list
        box = np.asarray(self.experiment.box_array, dtype=np.float32)
        rank, size = parallel.world()
        engine.set_device(config.device_index)
        rdf_engine = engine.RdfEngine(particles, box, np.float32(self.args.cutoff),
                                      self.args.number_of_bins,
                                      strict_reference_masks=self.strict_reference_masks)
        by_frames = size > 1 and (self.shard == "frames" or (
            self.shard == "auto" and rdf_engine.num_tiles < 8 * size))
        if by_frames:
            # this rank's frames, all tile pairs; the prefactor keeps the total count
            self.sample_configurations = parallel.split_frames(
                np.asarray(self.sample_configurations), rank, size)
            split_arr = [part for part in np.array_split(self.sample_configurations,
                                                         len(split_arr)) if len(part)]
            self.override_n_batches = max(1, len(split_arr))  # the loader re-plans its batches
            rank, size = 0, 1
        if len(self.sample_configurations) == 0:
            pass  # more ranks than frames: nothing to add on this one
        elif not self._accumulate_from_the_store(rdf_engine, rank, size):
            batch_ds = self.get_batch_dataset(subject_list=self.args.species,
                                              loop_array=split_arr, correct=True)
            for batch in batch_ds:
                positions = [np.ascontiguousarray(batch[key], dtype=np.float32)
                             for key in dict_keys]
                # tile sharding: every rank sees the same frames and evaluates its share
                # of the pair tiles
                rdf_engine.accumulate(positions, shard_rank=rank, shard_count=size)
        histograms = parallel.reduce_histograms(rdf_engine)  # the one exchange step
        rdf_engine.close()
        self.histograms = {key: histograms[i] for i, key in enumerate(self.key_list)}
        for i, key in enumerate(self.key_list):
            self.rdf[key] = histograms[i].astype(np.float64)
        self._calculate_radial_distribution_functions()
