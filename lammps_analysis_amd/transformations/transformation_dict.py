"""Which transformation can produce which property; used when a transformation's own
input is missing (mdsuite/transformations/transformation_dict.py:46-62 of the
reference).  A list is tried in order until one finds its inputs."""
from ..database.mdsuite_properties import mdsuite_properties as mdp
from .integrated_heat_current import IntegratedHeatCurrent
from .ionic_current import IonicCurrent
from .kinaci_integrated_heat_current import KinaciIntegratedHeatCurrent
from .momentum_flux import MomentumFlux
from .scale_coordinates import ScaleCoordinates
from .thermal_flux import ThermalFlux
from .translational_dipole_moment import TranslationalDipoleMoment
from .unwrap_coordinates import CoordinateUnwrapper
from .unwrap_via_indices import UnwrapViaIndices
from .velocity_from_positions import VelocityFromPositions
from .wrap_coordinates import CoordinateWrapper

property_to_transformation_dict = {
    mdp.integrated_heat_current: IntegratedHeatCurrent,
    mdp.ionic_current: IonicCurrent,
    mdp.kinaci_heat_current: KinaciIntegratedHeatCurrent,
    mdp.momentum_flux: MomentumFlux,
    mdp.positions: [ScaleCoordinates, CoordinateWrapper],
    mdp.thermal_flux: ThermalFlux,
    mdp.translational_dipole_moment: TranslationalDipoleMoment,
    mdp.unwrapped_positions: [UnwrapViaIndices, CoordinateUnwrapper],
    mdp.velocities_from_positions: VelocityFromPositions,
}
