"""Transformation base classes for derived system properties.

Follows the multi-species flow of mdsuite/transformations/transformations.py:66-647
of the reference: find every input property per species (a stored dataset, else
a per-species constant such as the charge), stream the trajectory in
memory-manager-sized batches, call ``transform_batch`` and append the result to
the ``Observables/<name>`` dataset of shape ``(1, T, k)``, stored as float32
(:289-291,315).  ``transform_batch`` of the concrete classes runs on the GPU.
"""
from __future__ import annotations

import abc
import collections.abc
import logging
import typing

import numpy as np

from .. import parallel
from ..database.data_manager import DataManager
from ..database.simulation_database import (Database, PropertyInfo, SpeciesInfo,  # noqa: F401
                                            open_database,
                                            TrajectoryChunkData)
from ..memory_management.memory_manager import MemoryManager
from ..utils.constants import DatasetKeys
from ..utils.meta_functions import join_path


class CannotFindTransformationError(Exception):
    pass


class CannotFindPropertyError(Exception):
    pass


class Transformations:
    def __init__(self, input_properties: typing.Iterable[PropertyInfo] = None,
                 output_property: PropertyInfo = None, scale_function=None,
                 dtype=np.float64):
        self._experiment = None
        self._database = None
        self.input_properties = input_properties
        self.output_property = output_property
        self.logger = logging.getLogger(__name__)
        self.scale_function = scale_function
        self.dtype = dtype
        self.batch_size: int = None
        self.n_batches: int = None
        self.remainder: int = None
        self.offset = 0
        self.data_manager: DataManager = None
        self.memory_manager: MemoryManager = None

    @property
    def database(self):
        if self._database is None:
            self._database = open_database(self.experiment.database_path / "database.hdf5")
        return self._database

    @property
    def experiment(self):
        return self._experiment

    @experiment.setter
    def experiment(self, value):
        self._experiment = value

    # ----------------------------------------------------------------- output

    def _save_output(self, data: np.ndarray, index: int, data_structure: dict):
        """Append one transformed batch ``[T_batch, k]`` (or ``[n, T_batch, k]``)."""
        key, val = list(data_structure.items())[0]
        sp_name, prop_name = key.rstrip("/").split("/")[-2:]
        n_particles = val.get("length")
        if n_particles is None:
            n_particles = len(val.get("indices"))
        data = np.asarray(data)
        if data.ndim == 2:
            data = data[np.newaxis, :, :]
        prop = PropertyInfo(name=prop_name, n_dims=len(val["columns"]))
        species = [SpeciesInfo(name=sp_name, properties=[prop], n_particles=n_particles)]
        chunk = TrajectoryChunkData(chunk_size=data.shape[1], species_list=species)
        chunk.add_data(data=np.swapaxes(data, 0, 1), config_idx=0, species_name=sp_name,
                       property_name=prop_name)
        self.database.add_data(chunk=chunk)

    def _prepare_monitors(self, data_path):
        self.memory_manager = MemoryManager(data_path=data_path, database=self.database,
                                            memory_fraction=0.5,
                                            scale_function=self.scale_function,
                                            offset=self.offset)
        self.batch_size, self.n_batches, self.remainder = \
            self.memory_manager.get_batch_size()
        self.data_manager = DataManager(data_path=data_path, data_range=1,
                                        database=self.database, batch_size=self.batch_size,
                                        n_batches=self.n_batches, remainder=self.remainder,
                                        offset=self.offset)

    def _prepare_database_entry(self, species: str, system_tensor=False):
        if system_tensor:
            output_length = 1
            path = join_path(DatasetKeys.OBSERVABLES, self.output_property.name)
        else:
            try:
                output_length = self.experiment.species[species].n_particles
            except KeyError:
                output_length = self.experiment.molecules[species].n_particles
            path = join_path(species, self.output_property.name)
        n_dims = self.output_property.n_dims
        if self.database.check_existence(path):
            old_shape = self.database.get_data_size(path)
            self.offset = old_shape[0]
            self.database.resize_datasets({path: (
                output_length, self.experiment.number_of_configurations - old_shape[0],
                n_dims)})
        else:
            self.database.add_dataset({path: (
                output_length, self.experiment.number_of_configurations, n_dims)})
        return {path: {"indices": np.s_[:], "columns": list(range(n_dims)),
                       "length": output_length}}

    # ----------------------------------------------------------------- inputs

    def find_property_per_config(self, sp_name, prop):
        would_be_path = join_path(sp_name, prop.name)
        return would_be_path if self.database.check_existence(would_be_path) else None

    def find_property_single_val(self, sp_name, prop):
        try:
            species_info = self.experiment.species[sp_name]
        except KeyError:
            species_info = self.experiment.molecules[sp_name]
        per_sp_value = getattr(species_info, prop.name.lower(), None)
        if per_sp_value is not None:
            return per_sp_value
        return getattr(self.experiment, prop.name.lower(), None)

    def get_prop_through_transformation(self, sp_name, prop):
        """Produce a missing input property by running the transformation that yields
        it (transformations.py:352-388 of the reference)."""
        from .transformation_dict import property_to_transformation_dict

        if prop not in property_to_transformation_dict:
            raise CannotFindTransformationError(
                f"was asked to get '{prop.name}' for '{sp_name}', but there is no "
                "transformation to get that property")
        trafo_cls = property_to_transformation_dict[prop]
        if not isinstance(trafo_cls, collections.abc.Iterable):
            self.experiment.cls_transformation_run(trafo_cls(), species=[sp_name])
        else:
            for trafo_class in trafo_cls:
                try:
                    self.experiment.cls_transformation_run(trafo_class(), species=[sp_name])
                except CannotFindPropertyError:
                    continue
                break
            else:
                raise CannotFindTransformationError(
                    f"was asked to get '{prop.name}' for '{sp_name}'. There are "
                    f"transformations to get this property ({trafo_cls}), but none of them "
                    "have the required data")
        return self.find_property_per_config(sp_name, prop)

    def get_generator_type_spec_and_const_data(self, species_names):
        """(paths that are streamed, {species: {property: constant array}})."""
        paths = {}
        const_input_data = {}
        for species_name in species_names:
            const_input_data[species_name] = {}
            for prop in self.input_properties:
                path = self.find_property_per_config(species_name, prop)
                if path is not None:
                    paths[str.encode(path)] = prop
                    continue
                val = self.find_property_single_val(species_name, prop)
                if val is None:
                    try:
                        path = self.get_prop_through_transformation(species_name, prop)
                    except CannotFindTransformationError:
                        path = None
                    if path is None:
                        raise CannotFindPropertyError(
                            f"While performing transformation "
                            f"'{self.output_property.name}': Property '{prop.name}' for "
                            f"species '{species_name}' cannot be found in the simulation "
                            "database nor in the simulation metadata, nor can it be "
                            "obtained by a transformation")
                    paths[str.encode(path)] = prop
                    continue
                if not isinstance(val, collections.abc.Iterable):
                    val = [val]
                val = np.asarray(val, dtype=self.dtype)[None, None, :]
                const_input_data[species_name][prop.name] = val
        return paths, const_input_data

    @abc.abstractmethod
    def run_transformation(self, species: typing.Iterable[str] = None):
        raise NotImplementedError


class SingleSpeciesTrafo(Transformations):
    """Transformations applied to each species separately
    (mdsuite/transformations/transformations.py:439-519 of the reference)."""

    def run_transformation(self, species: typing.Iterable[str] = None) -> None:
        if species is None:
            species = list(self.experiment.species.keys())
        for species_name in species:
            if self.database.check_existence(join_path(species_name, self.output_property.name)):
                self.logger.info(f"{self.output_property.name} already exists for "
                                 f"{species_name}, skipping transformation")
                continue
            # Inputs are resolved before the output dataset is laid out: a
            # transformation that cannot find its inputs then leaves no empty dataset
            # behind that would make the next candidate skip the species.
            paths, const_input_data = self.get_generator_type_spec_and_const_data(
                [species_name])
            const_input_data = const_input_data[species_name]
            output_data_structure = self._prepare_database_entry(species_name)
            self._prepare_monitors(list(paths.keys()))
            batch_generator, batch_generator_args = self.data_manager.batch_generator()
            carryover = None
            for index, batch_dict in enumerate(batch_generator(*batch_generator_args)):
                batch_dict.pop(str.encode("data_size"))
                # the transformation only has to know about the property
                batch_wo_species = {key.decode().split("/")[-1]: val
                                    for key, val in batch_dict.items()}
                batch_wo_species.update(const_input_data)
                ret = self.transform_batch(batch_wo_species, carryover=carryover)
                if isinstance(ret, tuple):
                    transformed_batch, carryover = ret
                else:
                    transformed_batch = ret
                self._save_output(data=transformed_batch,
                                  data_structure=output_data_structure,
                                  index=index * self.batch_size)

    @abc.abstractmethod
    def transform_batch(self, batch: typing.Dict[str, np.ndarray],
                        carryover: typing.Any = None):
        raise NotImplementedError("transformation of a batch must be implemented")


class MultiSpeciesTrafo(Transformations):
    """Transformations that combine all species into one system property.

    Every output frame depends on that frame alone, so with several ranks (one per GPU)
    each rank reduces a contiguous share of every batch's frames and the shares are
    gathered in rank order before the writing rank stores the batch; a subclass whose
    frames depend on earlier ones (a ``carryover``) sets ``frame_shardable = False``."""

    frame_shardable = True

    def run_transformation(self, species: typing.Iterable[str] = None) -> None:
        if species is None:
            species = list(self.experiment.species.keys())
        if self.database.check_existence(
                join_path(self.output_property.name, self.output_property.name)):
            self.logger.info(f"{self.output_property.name} already exists for this "
                             "experiment, skipping transformation")
            return
        output_data_structure = self._prepare_database_entry(self.output_property.name,
                                                             system_tensor=True)
        paths, const_input_data = self.get_generator_type_spec_and_const_data(species)
        self._prepare_monitors(list(paths.keys()))
        batch_generator, batch_generator_args = self.data_manager.batch_generator()
        carryover = None
        for index, batch_dict in enumerate(batch_generator(*batch_generator_args)):
            batch_dict.pop(str.encode("data_size"))
            hierarchical = {sp_name: {} for sp_name in species}
            rank, size = parallel.world()
            shard = size > 1 and self.frame_shardable
            begin = end = None
            for key, val in batch_dict.items():
                sp_name, prop_name = key.decode().split("/")
                if shard:
                    # this rank's frames of the batch (a view of the stored array)
                    begin, end = parallel.split_range(val.shape[1], rank, size)
                    val = val[:, begin:end]
                hierarchical[sp_name][prop_name] = val
            for sp_name in hierarchical:
                hierarchical[sp_name].update(const_input_data[sp_name])
            if shard:
                if end > begin:
                    part = self.transform_batch(hierarchical, carryover=None)
                    if isinstance(part, tuple):
                        raise RuntimeError(f"{type(self).__name__} carries state between "
                                           "batches and cannot shard frames over ranks")
                    part = np.asarray(part, dtype=np.float64)
                else:
                    part = np.zeros((0, self.output_property.n_dims))
                transformed_batch = np.concatenate(parallel.all_gather_objects(part), axis=0)
            else:
                ret = self.transform_batch(hierarchical, carryover=carryover)
                if isinstance(ret, tuple):
                    transformed_batch, carryover = ret
                else:
                    transformed_batch = ret
            self._save_output(data=transformed_batch, data_structure=output_data_structure,
                              index=index * self.batch_size)

    @abc.abstractmethod
    def transform_batch(self, batch: typing.Dict[str, typing.Dict[str, np.ndarray]],
                        carryover: typing.Any = None):
        raise NotImplementedError("transformation of a batch must be implemented")


def weight_arguments(weight, n_atoms: int):
    """Split a charge-like input into (scalar, per-atom weights or None).

    The input is either a per-species constant shaped ``[1, 1, 1]`` or a dataset
    ``[n_atoms, T, 1]``; per-frame varying weights are not supported on the device.
    """
    weight = np.asarray(weight, dtype=np.float64)
    if weight.size == 1:
        return float(weight.reshape(-1)[0]), None
    if weight.ndim == 3 and weight.shape[0] == n_atoms:
        first = weight[:, :1, :]
        if not np.all(weight == first):
            raise NotImplementedError("time-dependent per-atom weights are not supported")
        return 1.0, np.ascontiguousarray(first.reshape(n_atoms))
    raise ValueError(f"cannot interpret a weight of shape {weight.shape}")
