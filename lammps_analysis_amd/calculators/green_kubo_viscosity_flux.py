"""Green-Kubo viscosity from a stored stress flux ("Stress_Visc")
(mdsuite/calculators/green_kubo_viscosity_flux.py:55-273 of the reference; see
green_kubo_flux_base for how the window operation is read).  The stored
autocorrelation is divided by its maximum, as the reference's averaging step does
(:172-181)."""
from ..database.mdsuite_properties import mdsuite_properties
from .green_kubo_flux_base import GreenKuboFluxCalculator


class GreenKuboViscosityFlux(GreenKuboFluxCalculator):
    result_name = "viscosity"
    normalise_acf = True

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.loaded_property = mdsuite_properties.stress_viscosity
        self.y_label = r"$$\text{JACF} / C^{2}\\cdot m^{2}/s^{2}$$"
        self.analysis_name = "Viscosity_Flux"

    def _calculate_prefactor(self):
        exp = self.experiment
        denominator = (3 * (self.args.data_range - 1) * exp.temperature
                       * exp.units.boltzmann)
        prefactor_units = (exp.units.pressure**2 * exp.units.volume * exp.units.time
                           / exp.units.energy)
        self.prefactor = (exp.volume / denominator) * prefactor_units
