"""Kirkwood-Buff integrals ``G(r) = 4 pi int (g(r') - 1) r'^2 dr'`` from a radial
distribution function (mdsuite/calculators/kirkwood_buff_integrals.py:47-206 of the
reference)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.integrate import cumulative_trapezoid

from ..database.scheme import Computation
from ..utils.meta_functions import apply_savgol_filter
from ._rdf_consumer import RdfConsumer
from .calculator import call


@dataclass
class Args:
    savgol_order: int
    savgol_window_length: int
    number_of_bins: int
    number_of_configurations: int
    cutoff: float


class KirkwoodBuffIntegral(RdfConsumer):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.rdf_data = None
        self.database_group = "Kirkwood_Buff_Integral"
        self.x_label = r"$$ \text{r} / nm$$"
        self.y_label = r"$$\text{G}(\mathbf{r})$$"
        self.analysis_name = "Kirkwood-Buff_Integral"
        self.result_series_keys = ["r", "kb_integral"]
        self.data_range = 1
        self.post_generation = True

    @call
    def __call__(self, rdf_data: Computation = None, plot=True, savgol_order: int = 2,
                 savgol_window_length: int = 17):
        bins, cutoff, configurations = self._take_rdf(rdf_data)
        self.plot = plot
        self.args = Args(savgol_order=savgol_order, savgol_window_length=savgol_window_length,
                         number_of_bins=bins, cutoff=cutoff,
                         number_of_configurations=configurations)

    def _calculate_kb_integral(self, radii_data: np.ndarray, rdf_data: np.ndarray):
        filtered_data = apply_savgol_filter(rdf_data, order=self.args.savgol_order,
                                            window_length=self.args.savgol_window_length)
        integral_data = cumulative_trapezoid(
            y=(filtered_data[1:] - 1) * (radii_data[1:]) ** 2, x=radii_data[1:])
        return 4 * np.pi * integral_data

    def run_calculator(self):
        for selected_species, vals in self.rdf_data.data_dict.items():
            selected_species = selected_species.split("_")
            radii = np.array(vals["x"]).astype(float)[1:]
            rdf = np.array(vals["y"]).astype(float)[1:]
            kb_integral = self._calculate_kb_integral(radii_data=radii, rdf_data=rdf)
            data = {self.result_series_keys[0]: radii[1:].tolist(),
                    self.result_series_keys[1]: kb_integral.tolist()}
            self.queue_data(data=data, subjects=selected_species)
