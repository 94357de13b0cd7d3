"""Green-Kubo viscosity from the momentum flux
(mdsuite/calculators/green_kubo_viscosity.py:55-275 of the reference; see
green_kubo_flux_base for how the window operation is read)."""
from ..database.mdsuite_properties import mdsuite_properties
from .green_kubo_flux_base import GreenKuboFluxCalculator


class GreenKuboViscosity(GreenKuboFluxCalculator):
    result_name = "viscosity"

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.loaded_property = mdsuite_properties.momentum_flux
        self.y_label = r"\text{SACF} / C^{2}\cdot m^{2}/s^{2}$$"
        self.analysis_name = "Green_Kubo_Viscosity"

    def _calculate_prefactor(self):
        exp = self.experiment
        denominator = (3 * (self.args.data_range - 1) * exp.temperature
                       * exp.units.boltzmann * exp.volume)
        prefactor_units = (exp.units.pressure**2 * exp.units.volume * exp.units.time
                           / exp.units.energy)
        self.prefactor = (1 / denominator) * prefactor_units
