"""Einstein self-diffusion coefficients (mean-square displacement per species).

Host logic follows mdsuite/calculators/einstein_diffusion_coefficients.py:51-248 of
the reference, including the ``n_windows * (N + 1)`` normaliser (:184,244).  The
window loop runs on the GPU (``engine.msd_windows``).
"""
from __future__ import annotations

import logging
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

log = logging.getLogger(__name__)


@dataclass
class Args:
    data_range: int
    correlation_time: int
    atom_selection: np.s_
    tau_values: np.s_
    molecules: bool
    species: list
    fit_range: int


class EinsteinDiffusionCoefficients(TrajectoryCalculator, ABC):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.scale_function = {"linear": {"scale_factor": 150}}
        self.loaded_property = mdsuite_properties.unwrapped_positions
        self.x_label = r"$$\text{Time} / s$$"
        self.y_label = r"$$\text{MSD} / m^{2}$$"
        self.result_keys = ["diffusion_coefficient", "uncertainty", "gradient", "intercept"]
        self.result_series_keys = ["time", "msd", "gradients", "gradient_errors"]
        self.analysis_name = "Einstein Self-Diffusion Coefficients"
        self._dtype = np.float64
        self.msd_array = None

    @call
    def __call__(self, plot: bool = True, species: list = None, data_range: int = 100,
                 correlation_time: int = 1, atom_selection: np.s_ = np.s_[:],
                 molecules: bool = False, tau_values: Union[int, List, Any] = np.s_[:],
                 fit_range: int = -1):
        if species is None:
            species = list(self.experiment.molecules if molecules
                           else self.experiment.species)
        if fit_range == -1:
            fit_range = int(data_range - 1)
        self.args = Args(data_range=data_range, correlation_time=correlation_time,
                         atom_selection=atom_selection, tau_values=tau_values,
                         molecules=molecules, species=species, fit_range=fit_range)
        self.plot = plot
        self.system_property = False

    def fit_diff_coeff(self):
        self.msd_array /= self.count
        self.msd_array *= self.experiment.units.length**2
        self.time = self.time * self.experiment.units.time
        fit_values, covariance, gradients, gradient_errors = fit_einstein_curve(
            x_data=self.time, y_data=self.msd_array, fit_max_index=self.args.fit_range)
        error = np.sqrt(np.diag(covariance))[0]
        return {
            self.result_keys[0]: 1 / 6.0 * fit_values[0],
            self.result_keys[1]: 1 / 6.0 * error,
            self.result_keys[2]: fit_values[0],
            self.result_keys[3]: fit_values[1],
            self.result_series_keys[0]: self.time.tolist(),
            self.result_series_keys[1]: self.msd_array.tolist(),
            self.result_series_keys[2]: (np.array(gradients) / 6).tolist(),
            self.result_series_keys[3]: (np.array(gradient_errors) / 6).tolist(),
        }

    def run_calculator(self):
        self._run_dependency_check()
        engine.set_device(config.device_index)
        rank, size = parallel.world()
        for species in self.args.species:
            self.time = self._handle_tau_values()
            tau = np.asarray(self.args.tau_values, dtype=np.int32)
            dict_ref = str.encode("/".join([species, self.loaded_property.name]))
            self.msd_array = np.zeros(self.data_resolution)
            self.count = 0
            for batch in self.get_batch_dataset([species]):
                positions = batch[dict_ref]
                n_win = self.batch_windows(positions, batch[b"data_size"])
                if n_win == 0:
                    continue
                begin, end = parallel.split_range(n_win, rank, size)
                self.msd_array += engine.msd_windows(positions, self.args.correlation_time,
                                                     begin, end, tau, scale=1.0)
                self.count += n_win * (positions.shape[0] + 1)
            self.msd_array = parallel.all_reduce_sum(self.msd_array)
            self.queue_data(data=self.fit_diff_coeff(), subjects=[species])
