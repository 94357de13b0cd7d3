"""File-processor protocol (mdsuite/file_io/file_read.py:35-84 of the reference).

Text parsers (LAMMPS, extxyz, ...) are outside this engine's scope; any reader
that implements this protocol can feed ``Experiment.add_data``.
"""
from __future__ import annotations

import abc
import typing

from ..database.simulation_database import (SpeciesInfo, TrajectoryChunkData,
                                            TrajectoryMetadata)


class FileProcessor(abc.ABC):
    """Provides trajectory metadata and a generator of configuration chunks."""

    def __init__(self):
        self._metadata = None

    @abc.abstractmethod
    def __str__(self) -> str:
        """A unique name, used to remember which inputs were already added."""

    @property
    def metadata(self) -> TrajectoryMetadata:
        if getattr(self, "_metadata", None) is None:
            self._metadata = self._get_metadata()
        return self._metadata

    @abc.abstractmethod
    def _get_metadata(self) -> TrajectoryMetadata:
        """Collect the metadata needed to lay out the database."""

    @abc.abstractmethod
    def get_configurations_generator(self) -> typing.Iterator[TrajectoryChunkData]:
        """Yield the trajectory in chunks."""


def assert_species_list_consistent(sp_list_0: typing.List[SpeciesInfo],
                                   sp_list_1: typing.List[SpeciesInfo]):
    """Both lists must describe the same species, in any order."""
    if len(sp_list_0) != len(sp_list_1):
        raise ValueError("species lists differ in length")
    for sp_info_data in sp_list_0:
        matches = [sp for sp in sp_list_1 if sp.name == sp_info_data.name]
        if len(matches) != 1 or not (sp_info_data == matches[0]):
            raise ValueError(f"species '{sp_info_data.name}' is inconsistent between the "
                             "data and the metadata")
