"""Transformations that produce the system series the Green-Kubo / Einstein-Helfand
calculators correlate.  Each reduces per-atom arrays over the atom axis on the GPU."""
from .ionic_current import IonicCurrent
from .momentum_flux import MomentumFlux
from .thermal_flux import ThermalFlux
from .transformations import (CannotFindPropertyError, CannotFindTransformationError,
                              MultiSpeciesTrafo, Transformations)
from .translational_dipole_moment import TranslationalDipoleMoment

__all__ = ["IonicCurrent", "MomentumFlux", "ThermalFlux", "TranslationalDipoleMoment",
           "Transformations", "MultiSpeciesTrafo", "CannotFindPropertyError",
           "CannotFindTransformationError"]

# property name -> transformation that produces it (used by the calculators'
# dependency resolution, mdsuite/calculators/transformations_reference.py of the
# reference)
switcher_transformations = {
    "Translational_Dipole_Moment": "TranslationalDipoleMoment",
    "Ionic_Current": "IonicCurrent",
    "Momentum_Flux": "MomentumFlux",
    "Thermal_Flux": "ThermalFlux",
}
