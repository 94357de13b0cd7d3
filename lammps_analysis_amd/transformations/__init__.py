"""Transformations: the atom reductions that produce the system series the Green-Kubo /
Einstein-Helfand calculators correlate, and the per-atom coordinate transformations
(unwrap, wrap, scale, finite-difference velocities).  All run on the GPU."""
from .integrated_heat_current import IntegratedHeatCurrent
from .ionic_current import IonicCurrent
from .kinaci_integrated_heat_current import KinaciIntegratedHeatCurrent
from .momentum_flux import MomentumFlux
from .scale_coordinates import ScaleCoordinates
from .thermal_flux import ThermalFlux
from .transformations import (CannotFindPropertyError, CannotFindTransformationError,
                              MultiSpeciesTrafo, SingleSpeciesTrafo, Transformations)
from .translational_dipole_moment import TranslationalDipoleMoment
from .unwrap_coordinates import CoordinateUnwrapper
from .unwrap_via_indices import UnwrapViaIndices
from .velocity_from_positions import VelocityFromPositions
from .wrap_coordinates import CoordinateWrapper

__all__ = ["IntegratedHeatCurrent", "IonicCurrent", "KinaciIntegratedHeatCurrent", "MomentumFlux", "ThermalFlux", "TranslationalDipoleMoment",
           "CoordinateUnwrapper", "UnwrapViaIndices", "CoordinateWrapper", "ScaleCoordinates",
           "VelocityFromPositions", "Transformations", "SingleSpeciesTrafo",
           "MultiSpeciesTrafo", "CannotFindPropertyError", "CannotFindTransformationError"]

# property name -> transformation that produces it (used by the calculators'
# dependency resolution, mdsuite/calculators/transformations_reference.py of the
# reference)
switcher_transformations = {
    "Translational_Dipole_Moment": "TranslationalDipoleMoment",
    "Ionic_Current": "IonicCurrent",
    "Integrated_Heat_Current": "IntegratedHeatCurrent",
    "Kinaci_Heat_Current": "KinaciIntegratedHeatCurrent",
    "Momentum_Flux": "MomentumFlux",
    "Thermal_Flux": "ThermalFlux",
}
