"""Calculators of the accelerated path (same names as mdsuite.calculators)."""
from .angular_distribution_function import AngularDistributionFunction
from .coordination_number_calculation import CoordinationNumbers
from .einstein_diffusion_coefficients import EinsteinDiffusionCoefficients
from .einstein_distinct_diffusion_coefficients import EinsteinDistinctDiffusionCoefficients
from .einstein_helfand_ionic_conductivity import EinsteinHelfandIonicConductivity
from .einstein_helfand_thermal_conductivity import EinsteinHelfandThermalConductivity
from .einstein_helfand_thermal_kinaci import EinsteinHelfandThermalKinaci
from .green_kubo_distinct_diffusion_coefficients import \
    GreenKuboDistinctDiffusionCoefficients
from .green_kubo_ionic_conductivity import GreenKuboIonicConductivity
from .green_kubo_self_diffusion_coefficients import GreenKuboDiffusionCoefficients
from .green_kubo_thermal_conductivity import GreenKuboThermalConductivity
from .green_kubo_viscosity import GreenKuboViscosity
from .green_kubo_viscosity_flux import GreenKuboViscosityFlux
from .kirkwood_buff_integrals import KirkwoodBuffIntegral
from .nernst_einstein_ionic_conductivity import NernstEinsteinIonicConductivity
from .potential_of_mean_force import PotentialOfMeanForce
from .radial_distribution_function import RadialDistributionFunction

__all__ = ["AngularDistributionFunction", "CoordinationNumbers", "KirkwoodBuffIntegral",
           "PotentialOfMeanForce", "EinsteinDiffusionCoefficients", "EinsteinDistinctDiffusionCoefficients",
           "GreenKuboDistinctDiffusionCoefficients", "EinsteinHelfandIonicConductivity",
           "EinsteinHelfandThermalConductivity", "EinsteinHelfandThermalKinaci",
           "GreenKuboIonicConductivity", "GreenKuboDiffusionCoefficients",
           "GreenKuboThermalConductivity", "GreenKuboViscosity", "GreenKuboViscosityFlux",
           "RadialDistributionFunction", "NernstEinsteinIonicConductivity"]
