"""Green-Kubo self-diffusion coefficients (velocity autocorrelation per species).

Host logic follows
mdsuite/calculators/green_kubo_self_diffusion_coefficients.py:45-337 of the
reference, including its normaliser: ``count`` grows by the number of atoms and by
one per window (:196,334), so the averaged VACF is divided by ``n_windows * (N + 1)``.
"""
from __future__ import annotations

from abc import ABC
from dataclasses import dataclass
from typing import Any, List, Union

import numpy as np

from .. import engine, parallel
from ..database.mdsuite_properties import mdsuite_properties
from ..utils.calculator_helper_methods import cumulative_trapezoid
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


class GreenKuboDiffusionCoefficients(TrajectoryCalculator, ABC):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.loaded_property = mdsuite_properties.velocities
        self.scale_function = {"linear": {"scale_factor": 150}}
        self.x_label = r"$$\text{Time} / s$$"
        self.y_label = r"$$\text{VACF} / m^{2}s^{-2}$$"
        self.analysis_name = "Green Kubo Self-Diffusion Coefficients"
        self.result_keys = ["diffusion_coefficient", "uncertainty"]
        self.result_series_keys = ["time", "acf", "integral", "integral_uncertainty"]
        self._dtype = np.float64

    @call
    def __call__(self, plot: bool = True, species: list = None, data_range: int = 500,
                 correlation_time: int = 1, atom_selection=np.s_[:], molecules: bool = False,
                 tau_values: Union[int, List, Any] = np.s_[:],
                 integration_range: int = None):
        if species is None:
            species = list(self.experiment.molecules if molecules
                           else self.experiment.species)
        if integration_range is None:
            integration_range = data_range - 1
        self.args = Args(data_range=data_range, correlation_time=correlation_time,
                         atom_selection=atom_selection, tau_values=tau_values,
                         molecules=molecules, species=species,
                         integration_range=integration_range)
        self.plot = plot
        self.time = self._handle_tau_values() * self.experiment.units.time  # SI

    def check_input(self):
        self._run_dependency_check()

    def postprocessing(self, species: str):
        self.acf_array = self.acf_array / self.count
        sigma = cumulative_trapezoid(self.acf_array, x=self.time)
        sigma_sem = self.moments.std() / np.sqrt(self.moments.n)
        data = {
            self.result_keys[0]: [1 / 3 * sigma[self.args.integration_range - 1]],
            self.result_keys[1]: [1 / 3 * sigma_sem[self.args.integration_range - 1]],
            self.result_series_keys[0]: self.time.tolist(),
            self.result_series_keys[1]: self.acf_array.tolist(),
            self.result_series_keys[2]: sigma.tolist(),
            self.result_series_keys[3]: sigma_sem.tolist(),
        }
        self.queue_data(data=data, subjects=[species])

    def run_calculator(self):
        self.check_input()
        engine.set_device(config.device_index)
        rank, size = parallel.world()
        units = self.experiment.units
        scale = units.length**2 / units.time**2
        # the ACF is taken over the whole window; tau_values only label the times
        tau = np.arange(self.args.data_range, dtype=np.int32)
        for species in self.args.species:
            dict_ref = str.encode("/".join([species, self.loaded_property.name]))
            self.count = 0
            self.acf_array = np.zeros(self.data_resolution)
            self.moments = None
            for batch in self.get_batch_dataset([species]):
                velocities = batch[dict_ref]
                n_win = self.batch_windows(velocities, batch[b"data_size"])
                if n_win == 0:
                    continue
                n_atoms = velocities.shape[0]
                begin, end = parallel.split_range(n_win, rank, size)
                acf, moments = engine.acf_windows(
                    velocities, self.args.correlation_time, begin, end, tau, scale=scale,
                    integral_scale=1.0 / n_atoms, time=self.time)
                self.acf_array += acf
                self.count += n_win * (n_atoms + 1)
                self.moments = moments if self.moments is None else self.moments.merge(moments)
            self.acf_array = parallel.all_reduce_sum(self.acf_array)
            self.moments = parallel.merge_moments(self.moments)
            self.postprocessing(species)
