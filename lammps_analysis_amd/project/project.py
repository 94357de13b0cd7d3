"""Project: a directory holding experiments and their results.

Keeps the call surface of mdsuite/project/project.py:84-338 of the reference
(``Project(name, storage_path)``, ``add_experiment``, ``experiments``,
``active_experiments``, ``activate_experiments`` / ``disable_experiments``,
``add_data``, ``run``).  The reference keeps its state in an SQLite file; here
the experiment attributes and calculator results live in memory and are mirrored
to two JSON files in the project directory.
"""
from __future__ import annotations

import json
import logging
import pathlib
from datetime import datetime
from pathlib import Path
from typing import Dict, Union

from .. import parallel
from ..database import scheme as db
from ..experiment.experiment import Experiment
from ..experiment.run import RunComputation
from ..utils.helpers import NoneType
from ..utils.units import Units

log = logging.getLogger(__name__)


class dotdict(dict):
    """Dictionary whose keys are also attributes (``project.experiments.NaCl``)."""

    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


class ResultStore:
    """Calculator results of a project, searchable by (experiment, name, attributes)."""

    def __init__(self, path: pathlib.Path):
        self.path = path
        self._items = []
        if path.exists():
            try:
                with open(path) as fh:
                    raw = json.load(fh)
                for entry in raw:
                    results = []
                    for r in entry["results"]:
                        res = db.ComputationResult(r["data"], r["subjects"])
                        res.key = r["key"]
                        results.append(res)
                    self._items.append(db.Computation(entry["name"], entry["experiment"],
                                                      entry["attributes"], results))
            except (json.JSONDecodeError, KeyError):
                log.warning("could not read %s; starting with an empty result store", path)

    def find(self, experiment_name: str, name: str, attributes: dict):
        wanted = json.loads(json.dumps(attributes))
        for item in self._items:
            if (item.experiment_name == experiment_name and item.name == name
                    and item.computation_parameter == wanted):
                return item
        return None

    def add(self, computation: db.Computation) -> None:
        self._items.append(computation)
        self._save()

    def _save(self) -> None:
        if not parallel.is_writer():  # every rank holds the results in memory
            return
        raw = [{"name": c.name, "experiment": c.experiment_name,
                "attributes": c.computation_parameter,
                "results": [r.to_json() for r in c.computation_results]}
               for c in self._items]
        tmp = self.path.with_suffix(".tmp")
        with open(tmp, "w") as fh:
            fh.write(json.dumps(raw))  # one-shot: the C encoder (json.dump streams in Python)
        tmp.replace(self.path)


class Project:
    def __init__(self, name: str = None, storage_path: Union[str, Path] = "./",
                 description: str = None):
        self.name = "MDSuite_Project" if name is None else name
        self.storage_path = Path(storage_path).as_posix()
        self._experiments = {}
        self.project_dir = Path(f"{self.storage_path}/{self.name}")
        self.project_dir.mkdir(parents=True, exist_ok=True)
        self.attach_file_logger()
        # ranks of a multi-GPU job share the directory: all of them read the state the
        # previous run left before the writing rank touches it
        parallel.barrier()
        self._state_file = self.project_dir / "project.json"
        self._state = {"description": description, "experiments": {}}
        if self._state_file.exists():
            with open(self._state_file) as fh:
                self._state = json.load(fh)
            if description is not None:
                self._state["description"] = description
            log.info("Loading the class state")
        else:
            log.info(f"Creating new project {self.name}")
        self.results = ResultStore(self.project_dir / "results.json")
        parallel.barrier()
        self.save_state()

    # ----------------------------------------------------------------- state

    @property
    def description(self):
        return self._state.get("description")

    @description.setter
    def description(self, value):
        self._state["description"] = value
        self.save_state()

    def experiment_state(self, name: str) -> dict:
        return self._state["experiments"].setdefault(name, {})

    def save_state(self) -> None:
        if not parallel.is_writer():
            return
        tmp = self._state_file.with_suffix(".tmp")
        with open(tmp, "w") as fh:
            json.dump(self._state, fh)
        tmp.replace(self._state_file)

    def attach_file_logger(self):
        logger = logging.getLogger("mdsuite")
        channel = logging.FileHandler(self.project_dir / "mdsuite.log")
        channel.setLevel(logging.DEBUG)
        channel.setFormatter(logging.Formatter(
            "%(asctime)s %(levelname)s (%(module)s): %(message)s"))
        logger.addHandler(channel)

    def __str__(self):
        return "\n".join(f"{i}.) {name}"
                         for i, name in enumerate(self._state["experiments"], start=1))

    # ------------------------------------------------------------ experiments

    def add_experiment(self, name: str = NoneType, timestep: float = None,
                       temperature: float = None, units: Union[str, Units] = None,
                       cluster_mode: bool = None, active: bool = True,
                       simulation_data=None, update_with_pubchempy: bool = False) -> Experiment:
        if name is NoneType:
            raise ValueError("Experiment name can not be empty! "
                             "Use None to automatically generate a unique name.")
        if name is None:
            name = f"Experiment_{datetime.now().strftime('%Y%m%d-%H%M%S')}"
        if name in self._state["experiments"]:
            log.info("This experiment already exists")
            self.load_experiments(name)
            return self.experiments[name]
        new_experiment = Experiment(project=self, name=name, time_step=timestep,
                                    temperature=temperature, units=units,
                                    cluster_mode=cluster_mode)
        new_experiment.active = active
        self._experiments[name] = new_experiment
        if simulation_data is not None:
            self.experiments[name].add_data(simulation_data,
                                            update_with_pubchempy=update_with_pubchempy)
        return self.experiments[name]

    def load_experiments(self, names: Union[str, list]):
        self.activate_experiments(names)

    def activate_experiments(self, names: Union[str, list]):
        for name in ([names] if isinstance(names, str) else names):
            self.experiments[name].active = True

    def disable_experiments(self, names: Union[str, list]):
        for name in ([names] if isinstance(names, str) else names):
            self.experiments[name].active = False

    def add_data(self, data_sets: dict):
        for key, val in data_sets.items():
            self.experiments[key].add_data(val)

    @property
    def run(self) -> RunComputation:
        """Calculators bound to every active experiment."""
        return RunComputation(experiments=list(self.active_experiments.values()))

    @property
    def experiments(self) -> Dict[str, Experiment]:
        for name in self._state["experiments"]:
            if name not in self._experiments:
                self._experiments[name] = Experiment(project=self, name=name)
        return dotdict(self._experiments)

    @property
    def active_experiments(self) -> Dict[str, Experiment]:
        return dotdict({k: v for k, v in self.experiments.items() if v.active})
