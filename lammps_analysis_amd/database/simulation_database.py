"""Trajectory data model and the simulation database.

The data-model classes (``PropertyInfo``, ``SpeciesInfo``, ``MoleculeInfo``,
``TrajectoryMetadata``, ``TrajectoryChunkData``) carry the same fields as
mdsuite/database/simulation_database.py:43-227 of the reference, so that ingest
code written against it keeps working.

``Database`` keeps the reference's method surface (:230-702) on a different
storage engine.  The reference stores every dataset in one gzip-chunked HDF5 file
as float32 with the layout ``[n_particles, n_configurations, n_dims]``; h5py is
not a dependency here, and a compressed container is the wrong source for a GPU
that wants to stream frame blocks.  Each dataset is therefore one raw
``.npy`` file (float32, same layout, memory-mapped) inside a directory, with a
small JSON index holding shapes and the append position.  Reads return views of
the memory map, so a batch goes from the page cache to the device without an
intermediate copy.
"""
from __future__ import annotations

import collections
import dataclasses
import json
import logging
import os
import pathlib
import threading
import typing
from typing import List

import functools

import numpy as np

from .. import parallel

log = logging.getLogger(__name__)


def _writer_only(method):
    """Run a method that changes the store on the writing rank only; the other ranks of a
    multi-GPU job wait until it is done (they all share one project directory)."""

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        return parallel.run_on_writer(lambda: method(self, *args, **kwargs))

    return wrapper

STORAGE_DTYPE = np.float32  # the reference's h5py datasets default to float32 (:491-497)

# Read-only memory maps stay open between loads.  Setting up and tearing down the page
# tables of a multi-gigabyte mapping costs tens of milliseconds per calculator call, and
# pages touched once need no fault the second time.  A shared mapping shows whatever is
# written to the file through any other mapping; it goes stale only when the file is
# replaced or resized, so the key holds the inode and the size.
_READ_MAPS: "collections.OrderedDict[tuple, np.memmap]" = collections.OrderedDict()
_READ_MAPS_LIMIT = 32
_READ_MAPS_LOCK = threading.Lock()


def _read_map(path: pathlib.Path) -> np.memmap:
    stat = os.stat(path)
    key = (str(path), stat.st_ino, stat.st_size)
    with _READ_MAPS_LOCK:
        mapped = _READ_MAPS.get(key)
        if mapped is not None:
            _READ_MAPS.move_to_end(key)
            return mapped
        for stale in [k for k in _READ_MAPS if k[0] == key[0]]:
            del _READ_MAPS[stale]
        mapped = np.load(path, mmap_mode="r")
        _READ_MAPS[key] = mapped
        while len(_READ_MAPS) > _READ_MAPS_LIMIT:
            _READ_MAPS.popitem(last=False)
        return mapped


def release_read_maps() -> None:
    """Forget the cached read-only mappings (a deleted store's files stay allocated as long
    as a mapping of them is alive)."""
    with _READ_MAPS_LOCK:
        _READ_MAPS.clear()


@dataclasses.dataclass(frozen=True)
class PropertyInfo:
    """Name and dimensionality of a trajectory property, e.g. ("Positions", 3)."""

    name: str
    n_dims: int


@dataclasses.dataclass
class SpeciesInfo:
    """A species: its name, particle count, recorded properties, mass and charge."""

    name: str
    n_particles: int
    properties: List[PropertyInfo]
    mass: float = None
    charge: float = 0

    def __eq__(self, other):
        if len(self.properties) != len(other.properties):
            return False
        same = (self.name == other.name and self.n_particles == other.n_particles
                and self.mass == other.mass and self.charge == other.charge)
        return same and all(a == b for a, b in zip(self.properties, other.properties))


@dataclasses.dataclass
class MoleculeInfo(SpeciesInfo):
    """A molecule species; ``groups`` maps molecule index -> {species: [atom ids]}."""

    groups: dict = None

    def __eq__(self, other):
        if self.groups != other.groups:
            return False
        return super().__eq__(other)


@dataclasses.dataclass
class TrajectoryMetadata:
    """What must be known about a trajectory before its database can be laid out."""

    n_configurations: int
    species_list: List[SpeciesInfo]
    box_l: list = None
    sample_rate: int = 1
    sample_step: float = None
    temperature: float = None
    simulation_data: dict = dataclasses.field(default_factory=dict)


class TrajectoryChunkData:
    """A block of configurations on its way from a file reader to the database.

    Data is held per species and property as ``[chunk_size, n_particles, n_dims]``
    (time first, as file readers produce it).
    """

    def __init__(self, species_list: List[SpeciesInfo], chunk_size: int):
        self.chunk_size = chunk_size
        self.species_list = species_list
        self._data = {
            sp.name: {prop.name: np.zeros((chunk_size, sp.n_particles, prop.n_dims))
                      for prop in sp.properties}
            for sp in species_list
        }

    def add_data(self, data: np.ndarray, config_idx, species_name, property_name):
        """Store ``data`` ``[n_configs, n_particles, n_dims]`` from ``config_idx`` on."""
        data = np.asarray(data)
        n_configs = len(data)
        self._data[species_name][property_name][config_idx:config_idx + n_configs, :, :] = data

    def get_data(self):
        return self._data


class Database:
    """Directory-of-``.npy`` store with the reference ``Database`` interface."""

    INDEX = "index.json"

    def __init__(self, path: typing.Union[str, pathlib.Path] = "database"):
        self.path = pathlib.Path(path).as_posix() if isinstance(path, pathlib.Path) else path
        self._root = pathlib.Path(self.path)

    # ------------------------------------------------------------------ index

    def _load_index(self) -> dict:
        index_file = self._root / self.INDEX
        if not index_file.exists():
            return {}
        with open(index_file) as fh:
            return json.load(fh)

    def _save_index(self, index: dict) -> None:
        self._root.mkdir(parents=True, exist_ok=True)
        tmp = self._root / (self.INDEX + ".tmp")
        with open(tmp, "w") as fh:
            json.dump(index, fh)
        tmp.replace(self._root / self.INDEX)

    def _file(self, dataset: str) -> pathlib.Path:
        return self._root / (dataset.strip("/").replace("/", "__") + ".npy")

    # --------------------------------------------------------------- structure

    @staticmethod
    def _build_path_input(structure: dict) -> dict:
        """{'Na': {'Forces': shape}} -> {'Na/Forces': shape}; flat entries pass through."""
        architecture = {}
        for group, value in structure.items():
            if isinstance(value, dict):
                for subgroup, shape in value.items():
                    architecture[f"{group}/{subgroup}"] = shape
            else:
                architecture[group] = value
        return architecture

    def database_exists(self) -> bool:
        return (self._root / self.INDEX).exists()

    def initialize_database(self, structure: dict):
        self.add_dataset(self._build_path_input(structure))  # collective, see _writer_only

    @_writer_only
    def add_dataset(self, architecture: dict):
        """Create zero-filled datasets ``{path: shape}`` (append position 0)."""
        index = self._load_index()
        for dataset, shape in architecture.items():
            if not isinstance(shape, tuple):
                raise TypeError("Invalid input for dataset generation")
            dataset = dataset.strip("/")
            shape = tuple(int(s) for s in shape)
            self._root.mkdir(parents=True, exist_ok=True)
            mm = np.lib.format.open_memmap(self._file(dataset), mode="w+",
                                           dtype=STORAGE_DTYPE, shape=shape)
            del mm
            index[dataset] = {"shape": list(shape), "starting_index": 0}
        self._save_index(index)

    @_writer_only
    def resize_datasets(self, structure: dict):
        """Grow datasets along the configuration axis by the given amounts; unknown
        datasets are created (simulation_database.py:380-418 of the reference)."""
        architecture = self._build_path_input(structure)
        index = self._load_index()
        for dataset, info in architecture.items():
            if not isinstance(info, tuple):
                raise TypeError("Invalid input for dataset generation")
            dataset = dataset.strip("/")
            if dataset not in index:
                Database.add_dataset.__wrapped__(self, {dataset: info})
                index = self._load_index()
                continue
            axis = 0 if len(info[:-1]) == 1 else 1
            old = np.load(self._file(dataset), mmap_mode="r")
            new_shape = list(old.shape)
            new_shape[axis] = old.shape[axis] + int(info[axis])
            tmp = self._file(dataset).with_suffix(".grow.npy")
            grown = np.lib.format.open_memmap(tmp, mode="w+", dtype=STORAGE_DTYPE,
                                              shape=tuple(new_shape))
            selector = [slice(None)] * len(new_shape)
            selector[axis] = slice(0, old.shape[axis])
            grown[tuple(selector)] = old
            del grown, old
            tmp.replace(self._file(dataset))
            index[dataset]["shape"] = new_shape
        self._save_index(index)

    # -------------------------------------------------------------------- I/O

    @_writer_only
    def add_data(self, chunk: TrajectoryChunkData):
        """Append a chunk at every dataset's append position, transposing from the
        chunk's time-first layout to the stored ``[particles, configurations, dims]``."""
        index = self._load_index()
        chunk_data = chunk.get_data()
        for sp_info in chunk.species_list:
            for prop_info in sp_info.properties:
                dataset = f"{sp_info.name}/{prop_info.name}"
                if dataset not in index:
                    raise KeyError(f"dataset '{dataset}' is not in the database")
                write_data = chunk_data[sp_info.name][prop_info.name]
                start = index[dataset]["starting_index"]
                stop = start + chunk.chunk_size
                mm = np.load(self._file(dataset), mmap_mode="r+")
                if mm.ndim == 2:
                    mm[start:stop, :] = write_data[:, 0, :]
                elif mm.ndim == 3:
                    mm[:, start:stop, :] = np.swapaxes(write_data, 0, 1)
                else:
                    raise ValueError("dataset shape must be either (n_part,n_config,n_dim) "
                                     "or (n_config, n_dim)")
                mm.flush()
                del mm
                index[dataset]["starting_index"] = stop
        self._save_index(index)

    def load_data(self, path_list: list = None, select_slice=np.s_[:],
                  dictionary: bool = False, scaling: list = None, d_size: int = None,
                  dtype=None):
        """Load slices of datasets: ``{path: array, b"data_size": d_size}``.

        Arrays keep the stored float32 (``dtype=None``): the reference widens them to
        float64 on load (:626), the kernels do that widening on the device instead,
        with identical values.  ``select_slice`` may be a dict ``{species: slice}``.
        """
        if scaling is None:
            scaling = [1 for _ in range(len(path_list))]
        data = {}
        for i, item in enumerate(path_list):
            name = item.decode() if isinstance(item, bytes) else item
            if isinstance(select_slice, dict):
                my_slice = select_slice[name.split("/")[0]]
            else:
                my_slice = select_slice
            mm = _read_map(self._file(name))
            if (isinstance(my_slice, tuple) and len(my_slice) >= 2
                    and all(isinstance(part, (list, np.ndarray)) for part in my_slice[:2])):
                # an atom list and a frame list select a block (every listed atom at every
                # listed frame), as h5py applies them in the reference; NumPy would pair
                # the two lists element by element
                arr = mm[my_slice[0]][(slice(None), my_slice[1]) + tuple(my_slice[2:])]
            else:
                arr = mm[my_slice]
            arr = np.asarray(arr) if dtype is None else np.asarray(arr, dtype=dtype)
            if scaling[i] != 1:
                arr = arr * scaling[i]
            data[item] = arr
        data[str.encode("data_size")] = d_size
        return data

    # -------------------------------------------------------------- inspection

    def check_existence(self, path: str) -> bool:
        """True if a dataset whose path ends with ``/path`` exists."""
        if isinstance(path, bytes):
            path = path.decode()
        suffix = "/" + path.strip("/")
        return any(("/" + name).endswith(suffix) for name in self._load_index())

    def get_data_size(self, data_path: str) -> tuple:
        """(shape[0], shape[1], n_bytes) of a dataset."""
        if isinstance(data_path, bytes):
            data_path = data_path.decode()
        entry = self._load_index()[data_path.strip("/")]
        shape = entry["shape"]
        n_bytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(STORAGE_DTYPE).itemsize
        return shape[0], shape[1], n_bytes

    def get_memory_information(self) -> dict:
        out = {}
        for name, entry in self._load_index().items():
            out[name] = int(np.prod(entry["shape"], dtype=np.int64)) * 4
        return out

    def get_database_summary(self):
        return sorted({name.split("/")[0] for name in self._load_index()})

    @_writer_only
    def change_key_names(self, mapping: dict):
        """Rename top-level groups."""
        index = self._load_index()
        for name in list(index):
            group, _, rest = name.partition("/")
            if group in mapping:
                new_name = f"{mapping[group]}/{rest}" if rest else mapping[group]
                self._file(name).replace(self._file(new_name))
                index[new_name] = index.pop(name)
        self._save_index(index)


# ------------------------------------------------------------ HDF5 (optional)


def _h5py():
    """The h5py module, or an ImportError that says what it is needed for."""
    try:
        import h5py
    except ImportError as exc:
        raise ImportError(
            "reading an MDSuite database.hdf5 needs the optional dependency h5py; "
            "without it, ingest the trajectory again (Experiment.add_data), which writes "
            "the directory-of-.npy store") from exc
    return h5py


class Hdf5Database(Database):
    """Read side of a ``database.hdf5`` written by the reference.

    Layout contract (mdsuite/database/simulation_database.py:449-499, 594-639 of the
    reference): one group per species (or ``Observables``), one dataset per property,
    shape ``[n_particles, n_configurations, n_dims]``, float32, gzip-chunked.  Slices are
    read through h5py into contiguous float32 arrays, which the engine then uploads
    through its page-locked staging like any other host array; an atom list combined with
    a frame list is applied in two steps, as the reference does on h5py's ``TypeError``.

    The file itself is never written.  Datasets that transformations derive
    (``Observables/Ionic_Current`` ...) go to a directory-of-.npy store next to it
    (``<file>.mdx``) and are found there first on later reads.
    """

    def __init__(self, path: typing.Union[str, pathlib.Path]):
        super().__init__(path)
        self._h5_path = pathlib.Path(path)
        self._root = pathlib.Path(str(path) + ".mdx")
        self._h5_shapes = None

    def _h5_index(self) -> dict:
        if self._h5_shapes is None:
            shapes = {}
            with _h5py().File(self._h5_path, "r") as handle:
                def visit(name, node):
                    if hasattr(node, "shape") and hasattr(node, "dtype"):
                        shapes[name.strip("/")] = tuple(int(s) for s in node.shape)
                handle.visititems(visit)
            self._h5_shapes = shapes
        return self._h5_shapes

    def database_exists(self) -> bool:
        return self._h5_path.is_file()

    def _in_sidecar(self, name: str) -> bool:
        return name.strip("/") in self._load_index()

    def load_data(self, path_list: list = None, select_slice=np.s_[:],
                  dictionary: bool = False, scaling: list = None, d_size: int = None,
                  dtype=None):
        if scaling is None:
            scaling = [1 for _ in range(len(path_list))]
        data = {}
        from_file = []
        for i, item in enumerate(path_list):
            name = item.decode() if isinstance(item, bytes) else item
            if self._in_sidecar(name):
                data[item] = super().load_data([item], select_slice=select_slice,
                                               scaling=[scaling[i]], dtype=dtype)[item]
            else:
                from_file.append((i, item, name))
        if from_file:
            with _h5py().File(self._h5_path, "r") as handle:
                for i, item, name in from_file:
                    if isinstance(select_slice, dict):
                        my_slice = select_slice[name.split("/")[0]]
                    else:
                        my_slice = select_slice
                    dataset = handle[name]
                    try:
                        arr = dataset[my_slice]
                    except TypeError:
                        arr = dataset[my_slice[0]][:, my_slice[1], :]
                    arr = np.ascontiguousarray(arr, dtype=STORAGE_DTYPE if dtype is None
                                               else dtype)
                    if scaling[i] != 1:
                        arr = arr * scaling[i]
                    data[item] = arr
        data[str.encode("data_size")] = d_size
        return data

    def check_existence(self, path: str) -> bool:
        if super().check_existence(path):
            return True
        if isinstance(path, bytes):
            path = path.decode()
        suffix = "/" + path.strip("/")
        return any(("/" + name).endswith(suffix) for name in self._h5_index())

    def get_data_size(self, data_path: str) -> tuple:
        if isinstance(data_path, bytes):
            data_path = data_path.decode()
        name = data_path.strip("/")
        if self._in_sidecar(name):
            return super().get_data_size(name)
        shape = self._h5_index()[name]
        n_bytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(STORAGE_DTYPE).itemsize
        return shape[0], shape[1], n_bytes

    def get_memory_information(self) -> dict:
        out = {name: int(np.prod(shape, dtype=np.int64)) * 4
               for name, shape in self._h5_index().items()}
        out.update(super().get_memory_information())
        return out

    def get_database_summary(self):
        groups = {name.split("/")[0] for name in self._h5_index()}
        return sorted(groups | set(super().get_database_summary()))

    def add_data(self, chunk: TrajectoryChunkData):
        for sp_info in chunk.species_list:
            for prop_info in sp_info.properties:
                name = f"{sp_info.name}/{prop_info.name}"
                if not self._in_sidecar(name) and name in self._h5_index():
                    raise PermissionError(
                        f"dataset '{name}' lives in {self._h5_path}, which is opened "
                        "read-only; derived datasets go to the .mdx store beside it")
        return super().add_data(chunk)


def open_database(path: typing.Union[str, pathlib.Path]) -> Database:
    """The store behind ``path``: an HDF5 file written by the reference (needs h5py) or
    this package's directory of ``.npy`` datasets (also for a path that does not exist
    yet)."""
    if pathlib.Path(path).is_file():
        return Hdf5Database(path)
    return Database(path)
