"""Calculator base class and the ``@call`` protocol.

``@call`` drives a calculator exactly like mdsuite/calculators/calculator.py:52-148
of the reference: for each experiment build a fresh instance, let ``__call__`` store
the user arguments, look for a stored computation with the same arguments and
experiment version, otherwise prepare / save arguments / run / save results, then
return the stored ``Computation`` (a dict ``{experiment name: Computation}`` when
called through a project).  Plotting is outside this engine: ``plot=`` is accepted
and ignored.
"""
from __future__ import annotations

import functools
import logging
from typing import Dict, List, Union

import numpy as np

from ..database import scheme as db
from ..database.calculator_database import CalculatorDatabase

log = logging.getLogger(__name__)


def call(func):
    """Decorator of ``Calculator.__call__`` (see module docstring)."""

    @functools.wraps(func)
    def inner(self, *args, **kwargs) -> Union[db.Computation, Dict[str, db.Computation]]:
        return_dict = self.experiment is None
        out = {}
        for experiment in self.experiments:
            cls = self.__class__(experiment=experiment)
            func(cls, *args, **kwargs)
            data = cls.get_computation_data()
            if data is None:
                cls.prepare_db_entry()
                cls.save_computation_args()
                cls.run_analysis()
                cls.save_db_data()
                # run_analysis may have replaced defaults inside args; restore the
                # user's values so that the lookup matches what was saved
                func(cls, *args, **kwargs)
                data = cls.get_computation_data()
            out[cls.experiment.name] = data
        return out if return_dict else out[self.experiment.name]

    return inner


class Calculator(CalculatorDatabase):
    """Attributes shared by all calculators (calculator.py:151-317 of the reference)."""

    def __init__(self, experiment=None, experiments: List = None):
        super().__init__(experiment)
        self.experiment = experiment
        self.experiments = experiments
        if self.experiment is not None:
            self.experiments = [self.experiment]
        self.plot = False
        self.result_keys = None
        self.result_series_keys = None
        self.analysis_name = None
        self.selected_species = None
        self.system_property = False
        self.experimental = False
        self.time = None
        self.prefactor = None
        self._dtype = np.float64
        self.plotter = None
        self.x_label = None
        self.y_label = None
        self.plot_array = []

    @property
    def dtype(self):
        return self._dtype

    def run_calculator(self):
        raise NotImplementedError

    def plot_data(self, data):
        """Plotting is not part of this engine."""

    def run_analysis(self):
        if self.experimental:
            log.warning("This is an experimental calculator. Please see the "
                        "documentation before using the results.")
        self.run_calculator()
