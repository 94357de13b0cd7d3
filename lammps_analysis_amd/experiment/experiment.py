"""Experiment: one trajectory with its metadata, storage and calculators.

Keeps the user-facing surface of mdsuite/experiment/experiment.py:89-641 of the
reference (constructor arguments, ``run``, ``add_data``, ``load_matrix``,
``set_charge`` / ``set_mass``, ``cls_transformation_run``, ``units_to_si``) and the
attribute names of mdsuite/database/experiment_database.py:178-433 (``species``,
``molecules``, ``box_array``, ``units``, ``version``, ``volume``, ...).  Attributes
are kept in the project's JSON state instead of SQLite.
"""
from __future__ import annotations

import copy
import dataclasses
import logging
import pathlib
import typing
from pathlib import Path
from typing import Dict, Union

import numpy as np

from ..database.simulation_database import (Database, MoleculeInfo, PropertyInfo,  # noqa: F401
                                            open_database,
                                            SpeciesInfo, TrajectoryMetadata)
from ..file_io.file_read import FileProcessor
from ..utils import units as units_module
from ..utils.meta_functions import get_dimensionality, join_path
from ..utils.units import Units, units_dict
from .run import RunComputation

log = logging.getLogger(__name__)


def _get_processor(simulation_data):
    if isinstance(simulation_data, FileProcessor):
        return simulation_data
    if isinstance(simulation_data, (str, pathlib.Path)):
        raise ValueError(
            f"No text-file reader is part of this engine (got '{simulation_data}'). "
            "Pass a FileProcessor, e.g. file_io.script_input.ScriptInput.")
    raise ValueError("simulation_data must be an instance of file_io.file_read.FileProcessor."
                     f" Got '{type(simulation_data)}' instead")


def _species_list_to_architecture_dict(species_list, n_configurations):
    return {sp.name: {prop.name: (sp.n_particles, n_configurations, prop.n_dims)
                      for prop in sp.properties}
            for sp in species_list}


def _species_to_json(info: SpeciesInfo) -> dict:
    out = dataclasses.asdict(info)
    out.pop("name", None)
    out["properties"] = [[p["name"], p["n_dims"]] if isinstance(p, dict) else p
                         for p in out.get("properties", [])]
    return out


def _species_from_json(name: str, raw: dict, cls=SpeciesInfo):
    raw = dict(raw)
    props = [PropertyInfo(p[0], p[1]) if isinstance(p, (list, tuple)) else p
             for p in raw.pop("properties", [])]
    raw.pop("indices", None)
    return cls(name=name, properties=props, **raw)


class Experiment:
    def __init__(self, project, name, time_step=None, temperature=None,
                 units: Union[str, Units] = None, cluster_mode=False):
        if not name[0].isalpha():
            raise ValueError(
                f"Experiment name must start with a letter! Found '{name[0]}' instead.")
        self.project = project
        self.name = name
        self.storage_path = Path(project.storage_path, project.name).as_posix()
        self.cluster_mode = cluster_mode
        self._species = None
        self._molecules = None

        if self.number_of_configurations is None:
            self.number_of_configurations = 0
        self.temperature = temperature
        self.time_step = time_step
        if self.units is None:
            if units is None:
                units = units_module.REAL
            self.units = self.units_to_si(units)

        self.path = Path(self.storage_path, self.name)
        self.database_path = self.path
        self.figures_path = self.path / "figures"
        if not self.path.exists():
            log.info(f"Creating a new experiment ({self.name})!")
            self.path.mkdir(parents=True, exist_ok=True)
            self.figures_path.mkdir(exist_ok=True)

    # ------------------------------------------------------ persisted attributes

    def get_db(self, name: str, default=None):
        return self.project.experiment_state(self.name).get(name, default)

    def set_db(self, name: str, value):
        self.project.experiment_state(self.name)[name] = value
        self.project.save_state()

    def _simple(name, default=None):  # noqa: N805 - attribute factory
        def getter(self):
            return self.get_db(name, default)

        def setter(self, value):
            if value is None:
                return
            if isinstance(value, np.ndarray):
                value = value.tolist()
            elif isinstance(value, np.generic):
                value = value.item()
            self.set_db(name, value)

        return property(getter, setter)

    active = _simple("active", True)
    temperature = _simple("temperature")
    time_step = _simple("time_step")
    number_of_configurations = _simple("number_of_configurations")
    number_of_atoms = _simple("number_of_atoms")
    sample_rate = _simple("sample_rate")
    box_array = _simple("box_array")
    dimensions = _simple("dimensions")
    read_files = _simple("read_files", [])
    simulation_data = _simple("simulation_data", {})
    version = _simple("version", 0)
    property_groups = _simple("property_groups")
    del _simple

    @property
    def units(self) -> Union[Units, None]:
        raw = self.get_db("units")
        return None if raw is None else Units(**raw)

    @units.setter
    def units(self, value: Units):
        if value is not None:
            self.set_db("units", dataclasses.asdict(value))

    @property
    def species(self) -> Dict[str, SpeciesInfo]:
        if self._species is None:
            raw = self.get_db("species", {})
            self._species = {k: _species_from_json(k, v) for k, v in raw.items()}
        return self._species

    @species.setter
    def species(self, value: dict):
        if value is None:
            return
        if not isinstance(value, dict):
            raise ValueError("species must be a dict[str, SpeciesInfo] or dict[str, dict]")
        raw = {}
        for name, obj in value.items():
            if isinstance(obj, SpeciesInfo):
                raw[name] = _species_to_json(obj)
            else:
                obj = dict(obj)
                obj.pop("name", None)
                obj.pop("indices", None)
                obj["properties"] = [[p.name, p.n_dims] if isinstance(p, PropertyInfo) else p
                                     for p in obj.get("properties", [])]
                raw[name] = obj
        self._species = None
        self.set_db("species", raw)

    @property
    def molecules(self) -> Dict[str, MoleculeInfo]:
        if self._molecules is None:
            raw = self.get_db("molecules", {})
            self._molecules = {k: _species_from_json(k, v, MoleculeInfo)
                               for k, v in raw.items()}
        return self._molecules

    @molecules.setter
    def molecules(self, value):
        if value is None:
            return
        raw = {}
        for name, obj in value.items():
            raw[name] = _species_to_json(obj) if isinstance(obj, SpeciesInfo) else dict(obj)
        self._molecules = None
        self.set_db("molecules", raw)

    @property
    def volume(self):
        return np.prod(self.box_array)

    # ---------------------------------------------------------------- behaviour

    @property
    def run(self) -> RunComputation:
        """Calculators and transformations bound to this experiment."""
        return RunComputation(experiment=self)

    def __repr__(self):
        return f"exp_{self.name}"

    def cls_transformation_run(self, transformation, *args, **kwargs):
        transformation.experiment = self
        transformation.run_transformation(*args, **kwargs)

    @staticmethod
    def units_to_si(units_system) -> Units:
        if isinstance(units_system, Units):
            return units_system
        if isinstance(units_system, str):
            try:
                return units_dict[units_system]
            except KeyError:
                raise KeyError(f"The unit '{units_system}' is not implemented. "
                               f"The available units are: {list(units_dict)}")
        raise ValueError("units has to be of type Units or str, "
                         f"found {type(units_system)} instead")

    def set_charge(self, element: str, charge: float):
        species = self.species
        species[element].charge = [charge]
        self.species = species

    def set_mass(self, element: str, mass: float):
        species = self.species
        species[element].mass = mass
        self.species = species

    def add_data(self, simulation_data, force: bool = False,
                 update_with_pubchempy: bool = False):
        """Add one or several data sources (``FileProcessor`` instances)."""
        items = simulation_data if isinstance(simulation_data, list) else [simulation_data]
        for elem in items:
            self._add_data_from_file_processor(_get_processor(elem), force=force)

    def _add_data_from_file_processor(self, file_processor: FileProcessor, force: bool = False):
        if str(file_processor) in self.read_files and not force:
            log.info("This file has already been read, skipping this now. If this is not "
                     "desired, please add force=True to the command.")
            return
        database = open_database(self.database_path / "database.hdf5")
        metadata = file_processor.metadata
        architecture = _species_list_to_architecture_dict(metadata.species_list,
                                                          metadata.n_configurations)
        if not database.database_exists():
            self._store_metadata(metadata)
            database.initialize_database(architecture)
        else:
            database.resize_datasets(architecture)
        for batch in file_processor.get_configurations_generator():
            database.add_data(chunk=batch)
            self.number_of_configurations += batch.chunk_size
        self.version += 1
        self.read_files = self.read_files + [str(file_processor)]

    def load_matrix(self, property_name: str = None, species: typing.Iterable[str] = None,
                    select_slice=None, path: typing.Iterable[str] = None):
        database = open_database(self.database_path / "database.hdf5")
        if path is not None:
            return database.load_data(path_list=path, select_slice=select_slice)
        if species is None:
            species = list(self.species.keys())
        if select_slice is None:
            select_slice = np.s_[:]
        path_list = [join_path(item, property_name) for item in species]
        return database.load_data(path_list=path_list, select_slice=select_slice)

    def _store_metadata(self, metadata: TrajectoryMetadata):
        self.temperature = metadata.temperature
        self.box_array = metadata.box_l
        if self.box_array is not None:
            self.dimensions = get_dimensionality(self.box_array)
        self.sample_rate = metadata.sample_rate
        species_list = copy.deepcopy(metadata.species_list)
        self.species = {sp.name: sp for sp in species_list}
        self.property_groups = [p.name for p in species_list[0].properties]
        self.number_of_atoms = sum(sp.n_particles for sp in species_list)
