"""Green-Kubo distinct diffusion coefficients (velocity cross-correlations between
different atoms).

Host logic follows
mdsuite/calculators/green_kubo_distinct_diffusion_coefficients.py:53-362 of the
reference.  The reference averages ``jnp.correlate(a_i, b_j, "full")[D-1:]`` over all
atom pairs, O(n_a n_b D^2) per window; the correlation is bilinear, so the pair
average equals the correlation of the two species-mean velocity series, and the self
term (same species) is the mean per-atom autocorrelation.  The GPU evaluates exactly
that: one atom reduction per species, then ``mdx_ccf_windows``.

Kept from the reference: no 1/(D-m) lag normalisation, the prefactor
``length^2 / (time * (D-1))``, ``time`` without the unit factor in the per-window
trapezoid, and ``vacf`` accumulating across the species combinations (it is never
reset).  Accumulation is float64 where the reference's JAX default is float32, so
results agree to float32 rounding only (no offline test of the reference pins them).
"""
from __future__ import annotations

import itertools
from abc import ABC
from dataclasses import dataclass
from typing import Any, List, Union

import numpy as np

from .. import engine, parallel
from ..database.mdsuite_properties import mdsuite_properties
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
    integration_range: int


class GreenKuboDistinctDiffusionCoefficients(TrajectoryCalculator, ABC):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.scale_function = {"quadratic": {"inner_scale_factor": 5}}
        self.loaded_property = mdsuite_properties.velocities
        self.database_group = "Diffusion_Coefficients"
        self.x_label = r"$$\text{Time} / s$$"
        self.y_label = r"$$\text{VACF}  / m^{2}/s^{2}$$"
        self.analysis_name = "Green_Kubo_Distinct_Diffusion_Coefficients"
        self.experimental = True
        self.result_keys = ["diffusion_coefficient", "uncertainty"]
        self.result_series_keys = ["time", "vacf"]
        self._dtype = np.float64
        self.sigma = []

    @call
    def __call__(self, plot: bool = False, species: list = None, data_range: int = 500,
                 save: bool = True, correlation_time: int = 1,
                 tau_values: Union[int, List, Any] = np.s_[:], molecules: bool = False,
                 export: bool = False, atom_selection: dict = np.s_[:],
                 integration_range: int = None):
        if integration_range is None:
            integration_range = data_range
        self.args = Args(data_range=data_range, correlation_time=correlation_time,
                         atom_selection=atom_selection, tau_values=tau_values,
                         molecules=molecules, species=species,
                         integration_range=integration_range)
        self.plot = plot
        self.time = self._handle_tau_values()
        if len(self.args.tau_values) != self.args.data_range or \
                not np.array_equal(self.args.tau_values, np.arange(self.args.data_range)):
            raise NotImplementedError(
                "the GPU distinct-diffusion path correlates whole windows; sub-sampled "
                "tau_values are not supported")
        self.species = species
        self.vacf = np.zeros(self.args.data_range)
        if self.species is None:
            self.species = list(self.experiment.species)
        self.combinations = list(itertools.combinations_with_replacement(self.species, 2))

    def check_input(self):
        self._run_dependency_check()

    def _calculate_prefactor(self, species=None):
        numerator = self.experiment.units.length**2
        denominator = self.experiment.units.time * (self.args.data_range - 1)
        self.prefactor = numerator / denominator

    def _post_operation_processes(self, species=None):
        n = self.moments.n
        mean = self.prefactor * self.moments.mean[-1]
        sem = self.prefactor * self.moments.std()[-1] / np.sqrt(n)
        data = {
            self.result_keys[0]: float(mean),
            self.result_keys[1]: float(sem),
            self.result_series_keys[0]: self.time.tolist(),
            self.result_series_keys[1]: self.vacf.tolist(),
        }
        self.queue_data(data=data, subjects=list(species))

    def run_calculator(self):
        self.check_input()
        engine.set_device(config.device_index)
        rank, size = parallel.world()
        n_dims = self.loaded_property.n_dims
        for combination in self.combinations:
            species_values = list(combination)
            same_species = species_values[0] == species_values[1]
            dict_ref = [str.encode("/".join([sp, self.loaded_property.name]))
                        for sp in species_values]
            self.moments = None
            vacf = np.zeros(self.args.data_range)
            for batch in self.get_batch_dataset(species_values):
                vel_a, vel_b = batch[dict_ref[0]], batch[dict_ref[1]]
                n_win = self.batch_windows(vel_a, batch[b"data_size"])
                if n_win == 0:
                    continue
                begin, end = parallel.split_range(n_win, rank, size)
                # species-mean velocities: the pair average of a bilinear form
                mean_a = engine.weighted_atom_sum(vel_a, scalar_weight=1.0 / vel_a.shape[0])
                mean_b = mean_a if same_species else engine.weighted_atom_sum(
                    vel_b, scalar_weight=1.0 / vel_b.shape[0])
                self_atoms = np.ascontiguousarray(vel_a, dtype=np.float64) \
                    if same_species else None
                cf, moments = engine.ccf_windows(
                    mean_a, mean_b, self.args.correlation_time, begin, end,
                    self.args.data_range, cross_weight=1.0 / n_dims, self_atoms=self_atoms,
                    self_weight=1.0 / (n_dims * vel_a.shape[0]) if same_species else 0.0,
                    time=self.time)
                vacf += cf
                self.moments = moments if self.moments is None else self.moments.merge(moments)
            self.vacf += parallel.all_reduce_sum(vacf)
            self.moments = parallel.merge_moments(self.moments)
            self._calculate_prefactor(combination)
            self._post_operation_processes(combination)
            self.sigma = []
