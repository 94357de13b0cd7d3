"""Canonical property names and dimensions
(mdsuite/database/mdsuite_properties.py:54-84 of the reference)."""
from dataclasses import dataclass

from .simulation_database import PropertyInfo


@dataclass(frozen=True, init=False)
class MDSuiteProperties:
    temperature = PropertyInfo("Temperature", 1)
    time = PropertyInfo("Time", 1)
    thermal_flux = PropertyInfo("Thermal_Flux", 3)
    stress_viscosity = PropertyInfo("Stress_Visc", 3)
    momentum_flux = PropertyInfo("Momentum_Flux", 3)
    ionic_current = PropertyInfo("Ionic_Current", 3)
    translational_dipole_moment = PropertyInfo("Translational_Dipole_Moment", 3)
    positions = PropertyInfo("Positions", 3)
    scaled_positions = PropertyInfo("Scaled_Positions", 3)
    box_length = PropertyInfo("Box_Array", 3)
    unwrapped_positions = PropertyInfo("Unwrapped_Positions", 3)
    scaled_unwrapped_positions = PropertyInfo("Scaled_Unwrapped_Positions", 3)
    velocities = PropertyInfo("Velocities", 3)
    velocities_from_positions = PropertyInfo("Velocities_From_Positions", 3)
    momenta = PropertyInfo("Momenta", 3)
    forces = PropertyInfo("Forces", 3)
    box_images = PropertyInfo("Box_Images", 3)
    dipole_orientation_magnitude = PropertyInfo("Dipole_Orientation_Magnitude", 1)
    angular_velocity_spherical = PropertyInfo("Angular_Velocity_Spherical", 3)
    angular_velocity_non_spherical = PropertyInfo("Angular_Velocity_Non_Spherical", 3)
    torque = PropertyInfo("Torque", 3)
    integrated_heat_current = PropertyInfo("Integrated_Heat_Current", 3)
    kinaci_heat_current = PropertyInfo("Kinaci_Heat_Current", 3)
    charge = PropertyInfo("Charge", 1)
    energy = PropertyInfo("Energy", 1)
    kinetic_energy = PropertyInfo("Kinetic_Energy", 1)
    potential_energy = PropertyInfo("Potential_Energy", 1)
    stress = PropertyInfo("Stress", 6)
    time_step = PropertyInfo("Time_Step", 1)
    sample_rate = PropertyInfo("Sample_Rate", 1)


mdsuite_properties = MDSuiteProperties()
