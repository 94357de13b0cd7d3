"""Nernst-Einstein ionic conductivity from self-diffusion coefficients, optionally
corrected by the distinct ones.

Arithmetic of mdsuite/calculators/nernst_einstein_ionic_conductivity.py of the reference:
``sigma = N e^2 / (k_B T V) * sum_s D_s q_s^2 n_s / N`` (:172-205) and, corrected,
``+ sum_{ab} D_ab q_a q_b (n_a / N)(n_b / N)`` (:207-266), the uncertainties summed with
the same weights.  The reference reads the coefficients through
``experiment.export_property_data`` from its SQL store and picks the first entry per
analysis (:116-163); here the diffusion computations are handed in (or an Einstein
self-diffusion run with default arguments is made), as the radial-distribution consumers
do.  Host arithmetic only; the coefficients themselves come from the GPU calculators.
The reference has no enabled test for this file; parity unpinned.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ..database.scheme import Computation
from ..utils.units import boltzmann_constant, elementary_charge
from .calculator import Calculator, call


@dataclass
class Args:
    corrected: bool
    self_diffusion: str
    distinct_diffusion: str
    species: list


def _scalar(value) -> float:
    """Stored coefficients are floats (Einstein) or one-element lists (Green-Kubo)."""
    return float(np.ravel(np.asarray(value, dtype=float))[0])


class NernstEinsteinIonicConductivity(Calculator):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.post_generation = True
        self.database_group = "Ionic_Conductivity"
        self.analysis_name = "Nernst_Einstein_Ionic_Conductivity"
        self.result_keys = ["ionic_conductivity", "uncertainty"]
        self.self_diffusion = None
        self.distinct_diffusion = None

    @call
    def __call__(self, diffusion_data: Computation = None, distinct_diffusion_data: Computation = None,
                 corrected: bool = False, plot: bool = False, species: list = None):
        if diffusion_data is None:
            diffusion_data = self.experiment.run.EinsteinDiffusionCoefficients(plot=False)
        if corrected and distinct_diffusion_data is None:
            distinct_diffusion_data = self.experiment.run.EinsteinDistinctDiffusionCoefficients(
                plot=False)
        self.self_diffusion = diffusion_data
        self.distinct_diffusion = distinct_diffusion_data if corrected else None
        self.plot = plot
        if species is None:
            species = list(self.experiment.species)
        self.args = Args(corrected=bool(corrected), self_diffusion=diffusion_data.name,
                         distinct_diffusion=(distinct_diffusion_data.name
                                             if self.distinct_diffusion is not None else ""),
                         species=list(species))

    def _prefactor(self) -> float:
        numerator = self.experiment.number_of_atoms * elementary_charge**2
        denominator = (boltzmann_constant * self.experiment.temperature
                       * (self.experiment.volume * self.experiment.units.volume))
        return numerator / denominator

    def _fraction(self, name: str) -> float:
        return self.experiment.species[name].n_particles / self.experiment.number_of_atoms

    def _charge(self, name: str) -> float:
        return _scalar(self.experiment.species[name].charge)

    def run_calculator(self):
        conductivity = 0.0
        uncertainty = 0.0
        for name in self.args.species:
            if name not in self.self_diffusion.data_dict:
                raise ValueError(f"no self-diffusion coefficient for species '{name}' in "
                                 f"'{self.self_diffusion.name}'")
            entry = self.self_diffusion.data_dict[name]
            weight = self._charge(name) ** 2 * self._fraction(name)
            conductivity += _scalar(entry["diffusion_coefficient"]) * weight
            uncertainty += _scalar(entry["uncertainty"]) * weight
        if self.distinct_diffusion is not None:
            for pair_name, entry in self.distinct_diffusion.data_dict.items():
                a, b = pair_name.split("_")
                if a not in self.args.species or b not in self.args.species:
                    continue
                weight = (self._charge(a) * self._charge(b)
                          * self._fraction(a) * self._fraction(b))
                conductivity += _scalar(entry["diffusion_coefficient"]) * weight
                uncertainty += _scalar(entry["uncertainty"]) * weight
        prefactor = self._prefactor()
        self.queue_data(data={self.result_keys[0]: prefactor * conductivity,
                              self.result_keys[1]: prefactor * uncertainty},
                        subjects=["System"])
