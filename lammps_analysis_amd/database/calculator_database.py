"""Result caching of calculators.

Same protocol as mdsuite/database/calculator_database.py:70-248 of the reference
(``prepare_db_entry`` / ``get_computation_data`` / ``save_computation_args`` /
``queue_data`` / ``save_db_data``), backed by the project's in-memory result store
(persisted as JSON) instead of SQLite.  A stored computation is reused when the
calculator name, every field of its ``Args`` dataclass and the experiment version
match (:103-172).
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field, fields

from ..utils.meta_functions import is_jsonable
from . import scheme as db

log = logging.getLogger(__name__)


@dataclass
class ComputationResults:
    data: dict = field(default_factory=dict)
    subjects: dict = field(default_factory=list)


@dataclass
class Args:
    """Placeholder; every calculator defines its own ``Args`` dataclass."""


def conv_to_db(val):
    """Serialisable form of an argument value (slices etc. become strings)."""
    if is_jsonable(val):
        return val
    return str(val)


class CalculatorDatabase:
    def __init__(self, experiment):
        self.experiment = experiment
        self.db_computation: db.Computation = None
        self.database_group = None
        self.analysis_name = None
        self.load_data = None
        self.args = Args()
        self._queued_data = []
        self.db_computation_attributes = {}

    def _attribute_dict(self) -> dict:
        out = {f.name: conv_to_db(getattr(self.args, f.name)) for f in fields(self.args)}
        out["version"] = self.experiment.version
        return out

    def prepare_db_entry(self):
        self.db_computation = None
        self._queued_data = []

    def get_computation_data(self) -> db.Computation:
        """Stored computation with identical arguments and experiment version, if any."""
        return self.experiment.project.results.find(
            self.experiment.name, self.analysis_name, self._attribute_dict())

    def save_computation_args(self):
        self.db_computation_attributes = self._attribute_dict()

    def save_db_data(self):
        known = set(self.experiment.species) | set(self.experiment.molecules)
        results = [db.ComputationResult(q.data, q.subjects, known) for q in self._queued_data]
        self.db_computation = db.Computation(self.analysis_name, self.experiment.name,
                                             self.db_computation_attributes, results)
        self.experiment.project.results.add(self.db_computation)

    def queue_data(self, data, subjects):
        """Queue ``data`` (JSON-like dict) for the given subject names."""
        self._queued_data.append(ComputationResults(data=data, subjects=subjects))
