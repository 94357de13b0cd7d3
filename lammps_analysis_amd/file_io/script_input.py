"""In-memory ingest (mdsuite/file_io/script_input.py:13-45 of the reference): hand a
fully populated ``TrajectoryChunkData`` plus its metadata to an experiment."""
from __future__ import annotations

import typing

from ..database.simulation_database import TrajectoryChunkData, TrajectoryMetadata
from .file_read import FileProcessor, assert_species_list_consistent


class ScriptInput(FileProcessor):
    def __init__(self, data: TrajectoryChunkData, metadata: TrajectoryMetadata, name: str):
        super().__init__()
        self.data = data
        self.mdata = metadata
        self.name = name
        assert_species_list_consistent(data.species_list, metadata.species_list)
        if self.data.chunk_size != self.metadata.n_configurations:
            raise ValueError("Data must be provided in one chunk")

    def __str__(self):
        return self.name

    def _get_metadata(self) -> TrajectoryMetadata:
        return self.mdata

    def get_configurations_generator(self) -> typing.Iterator[TrajectoryChunkData]:
        yield self.data
