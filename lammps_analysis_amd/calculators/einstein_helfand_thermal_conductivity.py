"""Einstein-Helfand thermal conductivity (MSD of the integrated heat current).

Host logic follows
mdsuite/calculators/einstein_helfand_thermal_conductivity.py:45-261 of the reference:
the same windowed mean-square displacement as the ionic calculator with the prefactor
``units.energy / (units.length units.time units.temperature) / (V T k_B[units])`` and
``data_range`` defaulting to 500.  The reference's ``ensemble_operation`` indexes the
``(1, T, 3)`` series as if it were ``(T, 3)`` and has no enabled test; this calculator
computes the intended MSD of the system series.  Parity unpinned.
"""
from __future__ import annotations

from abc import ABC

import numpy as np

from ..database.mdsuite_properties import mdsuite_properties
from .calculator import call
from .einstein_helfand_ionic_conductivity import Args, EinsteinHelfandIonicConductivity


class EinsteinHelfandThermalConductivity(EinsteinHelfandIonicConductivity, ABC):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.loaded_property = mdsuite_properties.integrated_heat_current
        self.analysis_name = "Einstein Helfand Thermal Conductivity"
        self.result_keys = ["thermal_conductivity", "uncertainty"]

    @call
    def __call__(self, plot=True, data_range=500, correlation_time=1,
                 tau_values: np.s_ = np.s_[:], fit_range: int = -1):
        if fit_range == -1:
            fit_range = int(data_range - 1)
        self.args = Args(data_range=data_range, correlation_time=correlation_time,
                         tau_values=tau_values, atom_selection=np.s_[:],
                         fit_range=fit_range)
        self.plot = plot
        self.time = self._handle_tau_values()
        self.msd_array = np.zeros(self.data_resolution)

    def _calculate_prefactor(self):
        units = self.experiment.units
        denominator = self.experiment.volume * self.experiment.temperature * units.boltzmann
        units_change = units.energy / units.length / units.time / units.temperature
        self.prefactor = 1 / denominator * units_change
