"""Potential of mean force ``w(r) = -k_B T ln g(r)`` and its values at the minima between
the peaks (mdsuite/calculators/potential_of_mean_force.py:56-378 of the reference;
energies in eV through the reference's factor 6.242e8)."""
from __future__ import annotations

import logging
from dataclasses import dataclass

import numpy as np
from scipy.signal import find_peaks

from ..database.scheme import Computation
from ..utils.meta_functions import apply_savgol_filter, golden_section_search
from ..utils.units import boltzmann_constant
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


class PotentialOfMeanForce(RdfConsumer):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.rdf_data = None
        self.database_group = "Potential_Of_Mean_Force"
        self.x_label = r"$$r / nm$$"
        self.y_label = r"$$w^{(2)}(r) / eV$$"
        self.result_keys = []
        self.result_series_keys = ["r", "pomf"]
        self.analysis_name = "Potential_of_Mean_Force"
        self.post_generation = True

    @call
    def __call__(self, rdf_data: Computation = None, plot=True, savgol_order: int = 2,
                 savgol_window_length: int = 17, number_of_shells: int = 1):
        bins, cutoff, configurations = self._take_rdf(rdf_data)
        self.plot = plot
        self.args = Args(savgol_order=savgol_order, savgol_window_length=savgol_window_length,
                         number_of_bins=bins, cutoff=cutoff,
                         number_of_configurations=configurations,
                         number_of_shells=number_of_shells)
        self.result_keys = []
        for i in range(self.args.number_of_shells):
            self.result_keys.append(f"POMF_{i + 1}")
            self.result_keys.append(f"POMF_{i + 1}_error")

    def _calculate_potential_of_mean_force(self, rdf: np.ndarray) -> np.ndarray:
        with np.errstate(divide="ignore"):
            pomf = -1 * boltzmann_constant * self.experiment.temperature * np.log(rdf)
        return pomf * 6.242e8  # the reference's conversion to eV

    def get_pomf_peaks(self, pomf_data: np.ndarray) -> np.ndarray:
        filtered_data = apply_savgol_filter(pomf_data, order=self.args.savgol_order,
                                            window_length=self.args.savgol_window_length)
        peaks = find_peaks(filtered_data)[0]
        if len(peaks) < self.args.number_of_shells + 1:
            msg = ("Not enough peaks were detecting in the RDF to perform the desired "
                   "analysis. Try reducing the number of desired peaks or improving the "
                   "quality of the RDF provided.")
            log.error(msg)
            raise ValueError(msg)
        return peaks

    def _find_minimum(self, pomf_data: np.ndarray, radii_data: np.ndarray) -> dict:
        peaks = self.get_pomf_peaks(pomf_data)
        pomf_shells = {}
        for i in range(self.args.number_of_shells):
            pomf_shells[i] = np.zeros(2, dtype=int)
            pomf_radii_range = golden_section_search([radii_data, pomf_data],
                                                     radii_data[peaks[i + 1]],
                                                     radii_data[peaks[i]])
            for j in range(2):
                pomf_shells[i][j] = np.where(radii_data == pomf_radii_range[j])[0][0]
        return pomf_shells

    def _get_pomf_values(self, pomf: np.ndarray, radii: np.ndarray) -> dict:
        pomf_data = {}
        for key, val in self._find_minimum(pomf, radii).items():
            bounds = [pomf[val[0]], pomf[val[1]]]
            pomf_data[f"POMF_{int(key) + 1}"] = float(np.mean(bounds))
            pomf_data[f"POMF_{int(key) + 1}_error"] = float(np.std(bounds) / np.sqrt(2))
        return pomf_data

    def run_calculator(self):
        for selected_species, vals in self.rdf_data.data_dict.items():
            selected_species = selected_species.split("_")
            radii = np.array(vals["x"]).astype(float)[1:]
            rdf = np.array(vals["y"]).astype(float)[1:]
            pomf = self._calculate_potential_of_mean_force(rdf)
            pomf_data = self._get_pomf_values(pomf, radii)
            data = {self.result_series_keys[0]: radii[1:].tolist(),
                    self.result_series_keys[1]: pomf.tolist()}
            for item in self.result_keys:
                data[item] = pomf_data[item]
            self.queue_data(data=data, subjects=selected_species)
