"""Coordination numbers: the running integral ``4 pi rho int g(r) r^2 dr`` read off at
the minima between the peaks of a radial distribution function
(mdsuite/calculators/coordination_number_calculation.py:59-360 of the reference)."""
from __future__ import annotations

import logging
from dataclasses import dataclass

import numpy as np
from scipy.integrate import cumulative_trapezoid
from scipy.signal import find_peaks

from ..database.scheme import Computation
from ..utils.exceptions import CannotPerformThisAnalysis
from ..utils.meta_functions import apply_savgol_filter, golden_section_search
from ._rdf_consumer import RdfConsumer
from .calculator import call

log = logging.getLogger(__name__)


@dataclass
class Args:
    savgol_order: int
    savgol_window_length: int
    number_of_bins: int
    number_of_configurations: int
    cutoff: float
    number_of_shells: int


def _integrate_rdf(radii_data: np.ndarray, rdf_data: np.ndarray, density: float):
    integral_data = cumulative_trapezoid(y=radii_data[1:] ** 2 * rdf_data[1:],
                                         x=radii_data[1:])
    return 4 * np.pi * density * integral_data


class CoordinationNumbers(RdfConsumer):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.rdf_data = None
        self.volume = None
        self.post_generation = True
        self.database_group = "Coordination_Numbers"
        self.x_label = r"$$ \text{r} / nm$$"
        self.y_label = "CN"
        self.analysis_name = "Coordination_Numbers"
        self.result_keys = []
        self.result_series_keys = ["r", "cn"]

    @call
    def __call__(self, rdf_data: Computation = None, plot: bool = True,
                 savgol_order: int = 2, savgol_window_length: int = 17,
                 number_of_shells: int = 1):
        bins, cutoff, configurations = self._take_rdf(rdf_data)
        self.args = Args(savgol_order=savgol_order, savgol_window_length=savgol_window_length,
                         number_of_bins=bins, cutoff=cutoff,
                         number_of_configurations=configurations,
                         number_of_shells=number_of_shells)
        self.result_keys = []
        for i in range(self.args.number_of_shells):
            self.result_keys.append(f"CN_{i + 1}")
            self.result_keys.append(f"CN_{i + 1}_error")
        self._compute_nm_volume()
        self.plot = plot

    def _compute_nm_volume(self):
        volume_si = self.experiment.volume * self.experiment.units.volume
        self.volume = volume_si / 1e-9**3

    def _get_density(self, species: str) -> float:
        species = species.split("_")
        return self.experiment.species[species[0]].n_particles / self.volume

    def _get_rdf_peaks(self, rdf: np.ndarray) -> np.ndarray:
        filtered_data = apply_savgol_filter(rdf, order=self.args.savgol_order,
                                            window_length=self.args.savgol_window_length)
        peaks = find_peaks(filtered_data, height=1.0)[0]
        if len(peaks) < self.args.number_of_shells + 1:
            msg = ("Not enough peaks were detecting in the RDF to perform the desired "
                   "analysis. Try reducing the number of desired peaks or improving the "
                   "quality of the RDF provided.")
            log.error(msg)
            raise CannotPerformThisAnalysis(msg)
        return peaks

    def _find_minima(self, radii: np.ndarray, rdf: np.ndarray) -> dict:
        peaks = self._get_rdf_peaks(rdf)
        coordination_shells = {}
        for i in range(self.args.number_of_shells):
            coordination_shells[i] = np.zeros(2, dtype=int)
            cn_radii_range = golden_section_search([radii, rdf], radii[peaks[i + 1]],
                                                   radii[peaks[i]])
            for j in range(2):
                coordination_shells[i][j] = np.where(radii == cn_radii_range[j])[0][0]
        return coordination_shells

    def _get_coordination_numbers(self, integral_data, radii, rdf) -> dict:
        coordination_numbers = {}
        for key, val in self._find_minima(radii, rdf).items():
            bounds = [integral_data[val[0]], integral_data[val[1]]]
            coordination_numbers[f"CN_{int(key) + 1}"] = float(np.mean(bounds))
            coordination_numbers[f"CN_{int(key) + 1}_error"] = float(np.std(bounds) / np.sqrt(2))
        return coordination_numbers

    def run_calculator(self):
        for selected_species, vals in self.rdf_data.data_dict.items():
            radii = np.array(vals["x"]).astype(float)[1:]
            rdf = np.array(vals["y"]).astype(float)[1:]
            selected_species = selected_species.split("_")
            density = self._get_density(selected_species[0])
            integral_data = _integrate_rdf(radii, rdf, density)
            coordination_numbers = self._get_coordination_numbers(integral_data, radii, rdf)
            data = {self.result_series_keys[0]: radii[1:].tolist(),
                    self.result_series_keys[1]: integral_data.tolist()}
            for item in self.result_keys:
                data[item] = coordination_numbers[item]
            self.queue_data(data=data, subjects=selected_species)
