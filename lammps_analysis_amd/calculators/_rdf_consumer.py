"""Shared part of the calculators that post-process a radial distribution function on
the host: they take a finished ``RadialDistributionFunction`` computation (or run one
with default arguments, on the GPU) and work on its ``x`` / ``y`` series.  O(nbins)
NumPy / SciPy work, as in the reference."""
from __future__ import annotations

from ..database.scheme import Computation
from .calculator import Calculator


class RdfConsumer(Calculator):
    def _take_rdf(self, rdf_data):
        if isinstance(rdf_data, Computation):
            self.rdf_data = rdf_data
        else:
            self.rdf_data = self.experiment.run.RadialDistributionFunction(plot=False)
        parameter = self.rdf_data.computation_parameter
        return (parameter["number_of_bins"], parameter["cutoff"],
                parameter["number_of_configurations"])
