"""Green-Kubo thermal conductivity from the thermal flux
(mdsuite/calculators/green_kubo_thermal_conductivity.py:55-281 of the reference; see
green_kubo_flux_base for how the window operation is read)."""
from ..database.mdsuite_properties import mdsuite_properties
from .green_kubo_flux_base import GreenKuboFluxCalculator


class GreenKuboThermalConductivity(GreenKuboFluxCalculator):
    result_name = "computation_results"

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.loaded_property = mdsuite_properties.thermal_flux
        self.y_label = r"$$\text{JACF} / ($C^{2}\cdot m^{2}/s^{2}$$"
        self.analysis_name = "Green_Kubo_Thermal_Conductivity"

    def _calculate_prefactor(self):
        exp = self.experiment
        denominator = (3 * (self.args.data_range - 1) * exp.temperature**2
                       * exp.units.boltzmann * exp.volume)
        prefactor_units = exp.units.energy / exp.units.length / exp.units.time
        self.prefactor = (1 / denominator) * prefactor_units
