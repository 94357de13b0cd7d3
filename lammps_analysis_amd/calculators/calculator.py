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

    def stored_or_computed(calculator, args, kwargs) -> db.Computation:
        """The computation that belongs to the arguments of this call: the one on record,
        or a fresh run that is put on record first."""
        func(calculator, *args, **kwargs)
        on_record = calculator.get_computation_data()
        if on_record is not None:
            return on_record
        calculator.prepare_db_entry()
        calculator.save_computation_args()
        calculator.run_analysis()
        calculator.save_db_data()
        # a run may fill defaults into the stored arguments; the user's own values go back
        # in so that the look-up finds what was just saved
        func(calculator, *args, **kwargs)
        return calculator.get_computation_data()

    @functools.wraps(func)
    def inner(self, *args, **kwargs) -> Union[db.Computation, Dict[str, db.Computation]]:
        per_experiment = {}
        for experiment in self.experiments:
            fresh = type(self)(experiment=experiment)
            per_experiment[fresh.experiment.name] = stored_or_computed(fresh, args, kwargs)
        if self.experiment is None:          # called through a project: all of them
            return per_experiment
        return per_experiment[self.experiment.name]

    return inner


# attributes every calculator starts with (calculator.py:151-317 of the reference names them)
_CALCULATOR_DEFAULTS = dict(
    plot=False, plotter=None, x_label=None, y_label=None,
    analysis_name=None, result_keys=None, result_series_keys=None,
    selected_species=None, system_property=False, experimental=False,
    time=None, prefactor=None)


class Calculator(CalculatorDatabase):
    """What all calculators share: the experiment(s) they work on, the result keys, the
    ``@call`` protocol's hooks."""

    def __init__(self, experiment=None, experiments: List = None):
        super().__init__(experiment)
        self.experiment = experiment
        self.experiments = [experiment] if experiment is not None else experiments
        for name, value in _CALCULATOR_DEFAULTS.items():
            setattr(self, name, value)
        self.plot_array = []
        self._dtype = np.float64

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
