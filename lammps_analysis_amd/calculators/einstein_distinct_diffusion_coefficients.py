"""Einstein distinct diffusion coefficients (displacement cross-correlations between
different atoms).

Host logic follows
mdsuite/calculators/einstein_distinct_diffusion_coefficients.py:52-351 of the
reference.  Its pair average of ``msd_operation(a_i, b_j) = mean_d (a_i - a_i[0]) *
(b_j - b_j[0])`` is bilinear, so it equals the same product of the two species-mean
position series A, B; that product is obtained from two mean-square displacements by
polarisation, ``x y = ((x + y)^2 - (x - y)^2) / 4``, and the self term (same species)
is the per-atom mean-square displacement, all through ``mdx_msd_windows``.
Accumulation is float64 (JAX float32 in the reference); no offline test of the
reference pins the numbers.
"""
from __future__ import annotations

import itertools
from abc import ABC
from dataclasses import dataclass
from typing import Any, List, Union

import numpy as np

from .. import engine, parallel
from ..database.mdsuite_properties import mdsuite_properties
from ..utils.calculator_helper_methods import fit_einstein_curve
from ..utils.config import config
from .calculator import call
from .trajectory_calculator import TrajectoryCalculator


@dataclass
class Args:
    data_range: int
    correlation_time: int
    atom_selection: np.s_
    tau_values: np.s_
    molecules: bool
    species: list
    fit_range: int


class EinsteinDistinctDiffusionCoefficients(TrajectoryCalculator, ABC):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.scale_function = {"quadratic": {"inner_scale_factor": 10}}
        self.loaded_property = mdsuite_properties.unwrapped_positions
        self.database_group = "Diffusion_Coefficients"
        self.x_label = r"$$\text{Time} / s $$"
        self.y_label = r"$$\text{MSD} / m^{2}$$"
        self.analysis_name = "Einstein_Distinct_Diffusion_Coefficients"
        self.experimental = True
        self.result_keys = ["diffusion_coefficient", "uncertainty"]
        self.result_series_keys = ["time", "msd"]
        self.combinations = []
        self._dtype = np.float64

    @call
    def __call__(self, plot: bool = True, species: list = None, data_range: int = 500,
                 save: bool = True, correlation_time: int = 1,
                 tau_values: Union[int, List, Any] = np.s_[:], molecules: bool = False,
                 export: bool = False, atom_selection: dict = np.s_[:], fit_range: int = -1):
        if species is None:
            species = list(self.experiment.species)
        self.combinations = list(itertools.combinations_with_replacement(species, 2))
        self.plot = plot
        if fit_range == -1:
            fit_range = int(data_range - 1)
        self.args = Args(data_range=data_range, correlation_time=correlation_time,
                         atom_selection=atom_selection, tau_values=tau_values,
                         molecules=molecules, species=species, fit_range=fit_range)
        self.time = self._handle_tau_values() * self.experiment.units.time
        self.msd_array = np.zeros(self.data_resolution)

    def check_input(self):
        self._run_dependency_check()

    def _apply_averaging_factor(self):
        self.msd_array /= int(self.n_batches) * self.ensemble_loop

    def _post_operation_processes(self, species=None):
        self._apply_averaging_factor()
        self.msd_array *= self.experiment.units.length**2
        try:
            fit_values, covariance, _, _ = fit_einstein_curve(
                x_data=self.time, y_data=self.msd_array, fit_max_index=self.args.fit_range)
            sign = 1.0
        except ValueError:
            fit_values, covariance, _, _ = fit_einstein_curve(
                x_data=self.time, y_data=abs(self.msd_array),
                fit_max_index=self.args.fit_range)
            sign = -1.0
        error = np.sqrt(np.diag(covariance))[0]
        # the division by the dimension is part of the mean over dimensions, hence 1/2
        data = {
            self.result_keys[0]: sign * 1 / 2 * fit_values[0],
            self.result_keys[1]: 1 / 2 * error,
            self.result_series_keys[0]: self.time.tolist(),
            self.result_series_keys[1]: self.msd_array.tolist(),
        }
        self.queue_data(data=data, subjects=list(species))

    def run_calculator(self):
        self.check_input()
        engine.set_device(config.device_index)
        rank, size = parallel.world()
        n_dims = self.loaded_property.n_dims
        tau = np.asarray(self.args.tau_values, dtype=np.int32)
        ct = self.args.correlation_time
        for combination in self.combinations:
            species_values = list(combination)
            same_species = species_values[0] == species_values[1]
            dict_ref = [str.encode("/".join([sp, self.loaded_property.name]))
                        for sp in species_values]
            msd = np.zeros(self.data_resolution)
            for batch in self.get_batch_dataset(species_values):
                pos_a, pos_b = batch[dict_ref[0]], batch[dict_ref[1]]
                n_win = self.batch_windows(pos_a, batch[b"data_size"])
                if n_win == 0:
                    continue
                begin, end = parallel.split_range(n_win, rank, size)
                mean_a = engine.weighted_atom_sum(pos_a, scalar_weight=1.0 / pos_a.shape[0])
                if same_species:
                    cross = engine.msd_windows(mean_a[None], ct, begin, end, tau)
                    self_term = engine.msd_windows(pos_a, ct, begin, end, tau,
                                                   scale=1.0 / pos_a.shape[0])
                    msd += (cross - self_term) / n_dims
                else:
                    mean_b = engine.weighted_atom_sum(pos_b,
                                                      scalar_weight=1.0 / pos_b.shape[0])
                    plus = engine.msd_windows((mean_a + mean_b)[None], ct, begin, end, tau)
                    minus = engine.msd_windows((mean_a - mean_b)[None], ct, begin, end, tau)
                    msd += (plus - minus) / (4.0 * n_dims)
            self.msd_array = parallel.all_reduce_sum(msd)
            self._post_operation_processes(combination)
            self.msd_array = np.zeros(self.data_resolution)
