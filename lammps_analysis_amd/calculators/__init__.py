"""Calculators of the accelerated path (same names as mdsuite.calculators).

A calculator's module is imported when its name is first asked for, so that a run of the
pair histogram does not wait for the SciPy parts the correlation calculators pull in.
"""
import importlib

_MODULE_OF = {
    "AngularDistributionFunction": "angular_distribution_function",
    "CoordinationNumbers": "coordination_number_calculation",
    "EinsteinDiffusionCoefficients": "einstein_diffusion_coefficients",
    "EinsteinDistinctDiffusionCoefficients": "einstein_distinct_diffusion_coefficients",
    "EinsteinHelfandIonicConductivity": "einstein_helfand_ionic_conductivity",
    "EinsteinHelfandThermalConductivity": "einstein_helfand_thermal_conductivity",
    "EinsteinHelfandThermalKinaci": "einstein_helfand_thermal_kinaci",
    "GreenKuboDistinctDiffusionCoefficients": "green_kubo_distinct_diffusion_coefficients",
    "GreenKuboIonicConductivity": "green_kubo_ionic_conductivity",
    "GreenKuboDiffusionCoefficients": "green_kubo_self_diffusion_coefficients",
    "GreenKuboThermalConductivity": "green_kubo_thermal_conductivity",
    "GreenKuboViscosity": "green_kubo_viscosity",
    "GreenKuboViscosityFlux": "green_kubo_viscosity_flux",
    "KirkwoodBuffIntegral": "kirkwood_buff_integrals",
    "NernstEinsteinIonicConductivity": "nernst_einstein_ionic_conductivity",
    "PotentialOfMeanForce": "potential_of_mean_force",
    "RadialDistributionFunction": "radial_distribution_function",
}

__all__ = sorted(_MODULE_OF)


def __getattr__(name: str):
    try:
        module = _MODULE_OF[name]
    except KeyError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None
    value = getattr(importlib.import_module(f".{module}", __name__), name)
    globals()[name] = value
    return value


def __dir__():
    return sorted(set(globals()) | set(_MODULE_OF))
