"""Einstein-Helfand thermal conductivity from the Kinaci integrated heat current.

Host logic follows mdsuite/calculators/einstein_helfand_thermal_kinaci.py:43-267 of the
reference: the windowed mean-square displacement of the system series
``Kinaci_Heat_Current``, prefactor ``units.energy / (units.length units.time
units.temperature) / (V T k_B[units])`` (:156-177), average over ``n_batches *
ensemble_loop`` (:179-188), ``thermal_conductivity = slope / 6`` (:214-224),
``data_range`` defaulting to 500.  As in the other thermal Einstein-Helfand calculator the
reference's ``ensemble_operation`` (:190-205) indexes the ``(1, T, 3)`` series as if it
were ``(T, 3)`` and its test is disabled; this calculator computes the intended MSD of
the system series.  Parity unpinned.
"""
from __future__ import annotations

from abc import ABC

from ..database.mdsuite_properties import mdsuite_properties
from .einstein_helfand_thermal_conductivity import EinsteinHelfandThermalConductivity


class EinsteinHelfandThermalKinaci(EinsteinHelfandThermalConductivity, ABC):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.loaded_property = mdsuite_properties.kinaci_heat_current
        self.analysis_name = "Einstein Helfand Thermal Conductivity Kinaci"
