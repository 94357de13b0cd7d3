"""Result objects.

The reference keeps results in SQLite through an ORM (mdsuite/database/scheme.py);
that store is not on the compute path.  This module keeps only the object a
calculator call returns, ``Computation``, with the accessors user code relies on
(scheme.py:225-337): ``data_dict``, ``__getitem__``, ``keys()``,
``computation_parameter`` and ``data_range``.
"""
from __future__ import annotations

from collections import Counter
from typing import List


class ComputationResult:
    """Data of one subject group, e.g. the pair ("Na", "Cl") or ["System"]."""

    def __init__(self, data: dict, subjects: List[str], known_species=()):
        self.data = data
        self.subjects = list(subjects)
        # subjects that are not species of the experiment (e.g. "System") do not
        # contribute to the key, as in the reference's species association
        counted = Counter(s for s in self.subjects if s in known_species)
        names = []
        for name, count in counted.items():
            names += count * [name]
        self.key = "_".join(names) if names else "System"

    def to_json(self):
        return {"data": self.data, "subjects": self.subjects, "key": self.key}


class Computation:
    """A finished calculator run of one experiment."""

    def __init__(self, name: str, experiment_name: str, attributes: dict,
                 results: List[ComputationResult]):
        self.name = name
        self.experiment_name = experiment_name
        self._attributes = dict(attributes)
        self.computation_results = list(results)

    def __repr__(self):
        return f"Exp_{self.experiment_name}_{self.name}"

    @property
    def data_dict(self) -> dict:
        """{subject key: data}; species joined by "_" or "System"."""
        return {result.key: result.data for result in self.computation_results}

    def __getitem__(self, item):
        try:
            return self.data_dict[item]
        except KeyError:
            raise KeyError(
                f"Could not find {item} - available keys are {self.data_dict.keys()}")

    def keys(self) -> list:
        return list(self.data_dict.keys())

    @property
    def computation_parameter(self) -> dict:
        return dict(self._attributes)

    @property
    def data_range(self) -> int:
        value = self._attributes.get("data_range")
        return None if value is None else int(value)

    @property
    def subjects(self) -> list:
        out = []
        for result in self.computation_results:
            out += result.subjects
        return list(set(out))
