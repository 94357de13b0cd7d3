"""Shared body of the Green-Kubo calculators that correlate a stored flux series:
viscosity (momentum flux), thermal conductivity (thermal flux) and viscosity from a
pre-computed stress flux.

In the reference these three (mdsuite/calculators/green_kubo_viscosity.py:185-208,
green_kubo_thermal_conductivity.py:190-213, green_kubo_viscosity_flux.py:183-205)
call the autocorrelation along axis 0 of a ``(1, data_range, 3)`` window, i.e. over
a series of length one, which does not match the ``(1, T, 3)`` storage layout; their
tests are disabled.  This engine computes what the formulas intend and what the
tested ionic-conductivity calculator does: the windowed time autocorrelation, with
each calculator's own factors kept verbatim (``data_range *`` in the window
operation, ``1 / (data_range - 1)`` in the prefactor, the result taken from the
first window's integral and the "uncertainty" from the second window's).  No
reference result exists to compare against.

``reference_compat=True`` reproduces the files as written instead: the autocorrelation
of a length-one axis is the square, so every window contributes
``jacf_w[t] = data_range * sum_d x[t, d]**2`` (an array of shape ``(1, data_range)``),
``acf`` is the sum of those over the windows, and the result and its "uncertainty" are
the trapezoid integrals of the first and second window's ``jacf_w``.  That form needs
no correlation kernel (one pass of squares and a running sum over the series).
"""
from __future__ import annotations

from abc import ABC
from dataclasses import dataclass

import numpy as np

from .. import engine, parallel
from ..utils.config import config
from ..utils.constants import DatasetKeys
from .calculator import call
from .trajectory_calculator import TrajectoryCalculator


@dataclass
class Args:
    data_range: int
    correlation_time: int
    tau_values: np.s_
    atom_selection: np.s_
    integration_range: int
    reference_compat: bool = False


class GreenKuboFluxCalculator(TrajectoryCalculator, ABC):
    result_name = "computation_results"
    normalise_acf = False

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.scale_function = {"linear": {"scale_factor": 5}}
        self.system_property = True
        self.x_label = r"$$\text{Time} / s$$"
        self.prefactor = None
        self._dtype = np.float64

    @call
    def __call__(self, plot=False, data_range=500, tau_values: np.s_ = np.s_[:],
                 correlation_time: int = 1, integration_range: int = None,
                 reference_compat: bool = False):
        self.plot = plot
        self.reference_compat = bool(reference_compat)
        self.sigma = []
        if integration_range is None:
            integration_range = data_range
        self.args = Args(data_range=data_range, correlation_time=correlation_time,
                         tau_values=tau_values, atom_selection=np.s_[:],
                         integration_range=integration_range,
                         reference_compat=self.reference_compat)
        self.time = self._handle_tau_values()
        self.jacf = np.zeros(self.data_resolution)

    def check_input(self):
        self._run_dependency_check()

    def _calculate_prefactor(self):
        raise NotImplementedError

    def _window_integral(self, series, window: int, tau) -> float:
        """trapz(jacf[:integration_range], time[:integration_range]) of one window."""
        upper = min(self.args.integration_range, len(tau))
        if upper < 2:
            return 0.0
        _, moments = engine.acf_windows(series, self.args.correlation_time, window,
                                        window + 1, tau, scale=float(self.args.data_range),
                                        time=self.time)
        return float(moments.mean[upper - 2])

    def _run_as_written(self):
        """The reference's ensemble loop taken literally (see the module docstring)."""
        data_range, ct = self.args.data_range, self.args.correlation_time
        if self.args.integration_range != data_range:
            # np.trapz(jacf[:integration_range], x=time[:integration_range]) slices the
            # length-one axis of jacf, not its time axis: the shapes only agree for the
            # default integration range
            raise ValueError("operands could not be broadcast together: as written, the "
                             "reference integrates only with integration_range == data_range")
        dict_ref = str.encode("/".join([DatasetKeys.OBSERVABLES, self.loaded_property.name]))
        jacf = np.zeros((1, data_range))
        sigma = []
        trapezoid = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
        for batch in self.get_batch_dataset([DatasetKeys.OBSERVABLES]):
            series = np.asarray(batch[dict_ref], dtype=np.float64)
            n_win = self.batch_windows(series, batch[b"data_size"])
            if n_win == 0:
                continue
            squares = data_range * np.sum(series[0] * series[0], axis=-1)  # [T]
            running = np.concatenate([[0.0], np.cumsum(squares)])
            origins = np.arange(n_win) * ct
            for t in range(data_range):
                if ct == 1:
                    jacf[0, t] += running[t + n_win] - running[t]
                else:
                    jacf[0, t] += squares[origins + t].sum()
            for window in range(min(2, n_win)):
                if len(sigma) < 2:
                    part = squares[window * ct: window * ct + data_range]
                    sigma.append(trapezoid(part[None, :], x=self.time))
        result = self.prefactor * np.array(sigma)
        data = {
            self.result_name: result[0].tolist() if len(result) > 0 else None,
            "uncertainty": result[1].tolist() if len(result) > 1 else None,
            "time": self.time.tolist(),
            "acf": jacf.tolist(),
        }
        self.queue_data(data=data, subjects=["System"])

    def run_calculator(self):
        self.check_input()
        self._calculate_prefactor()
        if self.reference_compat:
            self._run_as_written()
            return
        engine.set_device(config.device_index)
        rank, size = parallel.world()
        tau = np.arange(self.args.data_range, dtype=np.int32)
        dict_ref = str.encode("/".join([DatasetKeys.OBSERVABLES, self.loaded_property.name]))
        self.jacf = np.zeros(self.args.data_range)
        self.sigma = []
        for batch in self.get_batch_dataset([DatasetKeys.OBSERVABLES]):
            series = batch[dict_ref]
            n_win = self.batch_windows(series, batch[b"data_size"])
            if n_win == 0:
                continue
            begin, end = parallel.split_range(n_win, rank, size)
            acf, _ = engine.acf_windows(series, self.args.correlation_time, begin, end, tau,
                                        scale=float(self.args.data_range))
            self.jacf += acf
            for window in range(min(2, n_win)):
                if len(self.sigma) < 2:
                    self.sigma.append(self._window_integral(series, window, tau))
        self.jacf = parallel.all_reduce_sum(self.jacf)
        if self.normalise_acf and np.max(self.jacf) != 0:
            self.jacf = self.jacf / np.max(self.jacf)
        result = self.prefactor * np.array(self.sigma)
        data = {
            self.result_name: float(result[0]) if len(result) > 0 else None,
            "uncertainty": float(result[1]) if len(result) > 1 else None,
            "time": self.time.tolist(),
            "acf": self.jacf.tolist(),
        }
        self.queue_data(data=data, subjects=["System"])
