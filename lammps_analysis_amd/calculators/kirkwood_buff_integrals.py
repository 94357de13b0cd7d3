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

    def running_integral(self, r: np.ndarray, g: np.ndarray) -> np.ndarray:
        """``4 pi * int_0^r (g - 1) r'^2 dr'`` on the grid ``r[1:]``.

        ``g`` is smoothed first (Savitzky-Golay, the two filter arguments of the call);
        the integrand's first point is left out, so the running trapezoid sum starts at
        ``r[1]`` and has ``len(r) - 2`` entries, as in the reference
        (mdsuite/calculators/kirkwood_buff_integrals.py:156-178)."""
        smooth = apply_savgol_filter(g, order=self.args.savgol_order,
                                     window_length=self.args.savgol_window_length)
        shell = r[1:] * r[1:]
        return 4.0 * np.pi * cumulative_trapezoid((smooth[1:] - 1.0) * shell, x=r[1:])

    def run_calculator(self):
        """One integral per species pair of the RDF.  The RDF's first bin (r = 0, where
        g is 0/0) is dropped before the filter sees it, and the integral's own first grid
        point after that, so ``r`` in the result starts at the RDF's third bin
        (mdsuite/calculators/kirkwood_buff_integrals.py:180-197)."""
        r_key, g_key = self.result_series_keys
        for pair_name, series in self.rdf_data.data_dict.items():
            r = np.asarray(series["x"], dtype=float)[1:]
            g = np.asarray(series["y"], dtype=float)[1:]
            self.queue_data(data={r_key: r[1:].tolist(),
                                  g_key: self.running_integral(r, g).tolist()},
                            subjects=pair_name.split("_"))
