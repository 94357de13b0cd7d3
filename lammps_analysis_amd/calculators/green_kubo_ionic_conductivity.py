"""Green-Kubo ionic conductivity.

Host logic follows mdsuite/calculators/green_kubo_ionic_conductivity.py:50-310 of
the reference.  The reference evaluates one FFT autocorrelation per window in a
Python loop and keeps every window's running integral in a list; here all windows
of a batch go to the GPU in one call (``engine.acf_windows``), which returns the
summed autocorrelation and streaming mean / variance of the running integrals.
"""
from __future__ import annotations

from abc import ABC
from dataclasses import dataclass

import numpy as np

from .. import engine, parallel
from ..database.mdsuite_properties import mdsuite_properties
from ..utils.calculator_helper_methods import cumulative_trapezoid
from ..utils.config import config
from ..utils.constants import DatasetKeys
from ..utils.units import boltzmann_constant, elementary_charge
from .calculator import call
from .trajectory_calculator import TrajectoryCalculator


@dataclass
class Args:
    data_range: int
    correlation_time: int
    tau_values: np.s_
    atom_selection: np.s_
    integration_range: int


class GreenKuboIonicConductivity(TrajectoryCalculator, ABC):
    """``experiment.run.GreenKuboIonicConductivity(data_range=500, correlation_time=10)``"""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.scale_function = {"linear": {"scale_factor": 5}}
        self.loaded_property = mdsuite_properties.ionic_current
        self.system_property = True
        self.x_label = r"$$\text{Time} / s$$"
        self.y_label = r"$$\text{JACF} / C^{2}\cdot m^{2}/s^{2}$$"
        self.analysis_name = "Green_Kubo_Ionic_Conductivity"
        self.result_keys = ["ionic_conductivity", "uncertainty"]
        self.result_series_keys = ["time", "acf", "integral", "integral_uncertainty"]
        self.prefactor = None
        self._dtype = np.float64

    @call
    def __call__(self, plot=True, data_range=500, correlation_time=1,
                 tau_values: np.s_ = np.s_[:], integration_range: int = None):
        self.plot = plot
        if integration_range is None:
            integration_range = data_range - 1
        self.args = Args(data_range=data_range, correlation_time=correlation_time,
                         tau_values=tau_values, atom_selection=np.s_[:],
                         integration_range=integration_range)
        self.time = self._handle_tau_values()
        self.jacf = np.zeros(self.data_resolution)

    def check_input(self):
        self._run_dependency_check()

    def _calculate_prefactor(self):
        numerator = (elementary_charge**2) * (self.experiment.units.length**2)
        denominator = (3 * boltzmann_constant * self.experiment.temperature
                       * self.experiment.volume * self.experiment.units.volume
                       * self.experiment.units.time)
        self.prefactor = numerator / denominator

    def _post_operation_processes(self):
        self.acf_array /= self.count
        sigma = cumulative_trapezoid(self.acf_array, x=self.time)
        sigma_sem = self.moments.std() / np.sqrt(self.moments.n)
        data = {
            self.result_keys[0]: [self.prefactor * sigma[self.args.integration_range - 1]],
            self.result_keys[1]: [self.prefactor * sigma_sem[self.args.integration_range - 1]],
            self.result_series_keys[0]: self.time.tolist(),
            self.result_series_keys[1]: self.acf_array.tolist(),
            self.result_series_keys[2]: sigma.tolist(),
            self.result_series_keys[3]: sigma_sem.tolist(),
        }
        self.queue_data(data=data, subjects=["System"])

    def run_calculator(self):
        self.check_input()
        self._calculate_prefactor()
        dict_ref = str.encode("/".join([DatasetKeys.OBSERVABLES, self.loaded_property.name]))
        engine.set_device(config.device_index)
        rank, size = parallel.world()
        tau = np.asarray(self.args.tau_values, dtype=np.int32)
        self.count = 0
        self.acf_array = np.zeros(self.data_resolution)
        self.moments = None
        for batch in self.get_batch_dataset([DatasetKeys.OBSERVABLES]):
            series = batch[dict_ref]
            n_win = self.batch_windows(series, batch[b"data_size"])
            if n_win == 0:
                continue
            begin, end = parallel.split_range(n_win, rank, size)  # time origins per GPU
            acf, moments = engine.acf_windows(series, self.args.correlation_time, begin, end,
                                              tau, scale=1.0, time=self.time)
            self.acf_array += acf
            self.count += n_win
            self.moments = moments if self.moments is None else self.moments.merge(moments)
        self.acf_array = parallel.all_reduce_sum(self.acf_array)
        self.moments = parallel.merge_moments(self.moments)
        self._post_operation_processes()
