"""NumPy-facing wrappers around the libmdx C-ABI.

These functions are the only place where the calculators touch the GPU.  Inputs
are NumPy arrays (host memory, staged by the library) or :class:`DeviceArray`
objects (data already resident in HBM); outputs are NumPy arrays.
"""
from __future__ import annotations

import ctypes
import dataclasses
from typing import Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import (MDX_DEVICE, MDX_F32, MDX_F64, MDX_HOST, MDX_RDF_LITERAL_KERNEL,
                   MDX_RDF_STRICT_REFERENCE_MASKS, MDX_RDF_TABLE_KERNEL, check)

ArrayLike = Union[np.ndarray, "DeviceArray"]


def set_device(index: int) -> None:
    from .utils.config import config
    if config.backend != "hip":
        raise ValueError(f"config.backend = {config.backend!r}: the only compute backend is "
                         "'hip' (libmdx on an AMD GPU); there is no CPU path")
    check(_lib.load().mdx_set_device(int(index)))


def synchronize() -> None:
    check(_lib.load().mdx_synchronize())


def release_cached_memory() -> None:
    """Return the scratch blocks the library keeps cached between calls to the driver."""
    check(_lib.load().mdx_release_cached_memory())


def device_info() -> dict:
    name = ctypes.create_string_buffer(256)
    cus = ctypes.c_int(0)
    mem = ctypes.c_uint64(0)
    check(_lib.load().mdx_device_info(name, 256, ctypes.byref(cus), ctypes.byref(mem)))
    return {"name": name.value.decode(), "compute_units": cus.value,
            "memory_bytes": mem.value}


def _dtype_code(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return MDX_F32
    if dtype == np.float64:
        return MDX_F64
    raise TypeError(f"unsupported dtype {dtype}; use float32 or float64")


class DeviceArray:
    """A C-contiguous array living in device memory (allocated by libmdx)."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        ptr = ctypes.c_void_p()
        check(_lib.load().mdx_malloc(ctypes.byref(ptr), self.nbytes))
        self.ptr = ptr.value

    @classmethod
    def from_numpy(cls, array: np.ndarray) -> "DeviceArray":
        array = np.ascontiguousarray(array)
        out = cls(array.shape, array.dtype)
        if out.nbytes:
            check(_lib.load().mdx_memcpy_h2d(out.ptr, array.ctypes.data, out.nbytes))
        return out

    def to_numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(_lib.load().mdx_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
        return out

    def fill_uniform(self, seed: int, lo: float, hi: float) -> None:
        """Deterministic synthetic fill (float32 only); see mdx_fill_uniform_f32."""
        if self.dtype != np.float32:
            raise TypeError("fill_uniform needs a float32 array")
        n = self.nbytes // 4
        check(_lib.load().mdx_fill_uniform_f32(self.ptr, n, int(seed), float(lo), float(hi)))

    def fill_langevin(self, seed: int, decay: float, kick: float) -> None:
        """Synthetic Ornstein-Uhlenbeck velocities along axis 1 of an ``[atoms, frames,
        k]`` float32 array; see mdx_fill_ou_f32."""
        if self.dtype != np.float32 or len(self.shape) != 3:
            raise TypeError("fill_langevin needs a float32 array of shape (atoms, frames, k)")
        check(_lib.load().mdx_fill_ou_f32(self.ptr, self.shape[0], self.shape[1],
                                          self.shape[2], int(seed), float(decay), float(kick)))

    def reshape(self, shape) -> "DeviceArray":
        """The same memory under another shape; the result does not own it and keeps
        this array alive."""
        shape = tuple(int(s) for s in shape)
        if int(np.prod(shape, dtype=np.int64)) * self.dtype.itemsize != self.nbytes:
            raise ValueError(f"cannot reshape {self.shape} to {shape}")
        view = object.__new__(DeviceArray)
        view.shape, view.dtype, view.nbytes, view.ptr = shape, self.dtype, self.nbytes, self.ptr
        view._base = self
        return view

    def free(self) -> None:
        if getattr(self, "_base", None) is not None:
            self.ptr = None
            self._base = None
            return
        if getattr(self, "ptr", None):
            _lib.load().mdx_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def uniform_fill_reference(n: int, seed: int, lo: float, hi: float, start: int = 0) -> np.ndarray:
    """NumPy restatement of ``mdx_fill_uniform_f32`` for elements start..start+n-1."""
    return uniform_fill_at(np.arange(start, start + n, dtype=np.uint64), seed, lo, hi)


def uniform_fill_at(index: np.ndarray, seed: int, lo: float, hi: float) -> np.ndarray:
    """Values ``mdx_fill_uniform_f32`` writes at the given flat element indices."""
    i = np.asarray(index, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (i + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    lo32 = np.float32(lo)
    span = np.float32(np.float32(hi) - lo32)
    return (lo32 + (span * u).astype(np.float32)).astype(np.float32)


def _as_input(x: ArrayLike, rows: bool = False, any_size: bool = False):
    """(pointer, dtype code, location, shape, keepalive) of an input array.  ``rows``: a
    frame range of a larger host array may be uploaded here, row by row (entry points with a
    single array argument)."""
    if isinstance(x, DeviceArray):
        return x.ptr, _dtype_code(x.dtype), MDX_DEVICE, x.shape, x
    arr = np.asarray(x)
    if arr.dtype not in (np.float32, np.float64):
        arr = arr.astype(np.float64)
    if rows and _is_frame_range(arr, any_size):
        # a frame range of a stored [atoms][frames][k] array (what the batch loader hands
        # over): its rows go to the device as they lie, without a contiguous host copy
        dev = DeviceArray(arr.shape, arr.dtype)
        check(_lib.load().mdx_memcpy_h2d_rows(dev.ptr, arr.ctypes.data, arr.shape[0],
                                              arr.shape[1] * arr.strides[1], arr.strides[0]))
        return dev.ptr, _dtype_code(arr.dtype), MDX_DEVICE, arr.shape, dev
    arr = np.ascontiguousarray(arr)
    return arr.ctypes.data, _dtype_code(arr.dtype), MDX_HOST, arr.shape, arr


def _as_inputs(arrays):
    """``_as_input`` for the arrays of one call, which must end up in one place: frame
    ranges are uploaded row by row only when every host array of the call is one."""
    host = [np.asarray(a) for a in arrays if not isinstance(a, DeviceArray)]
    rows = bool(host) and len(host) == len(arrays) and \
        sum(h.nbytes for h in host) >= _ROWS_UPLOAD_MIN_BYTES and \
        all(h.dtype in (np.float32, np.float64) and _is_frame_range(h, any_size=True)
            for h in host)
    return [_as_input(a, rows=rows, any_size=rows) for a in arrays]


_ROWS_UPLOAD_MIN_BYTES = 8 << 20


def _is_frame_range(arr: np.ndarray, any_size: bool = False) -> bool:
    """True for a large three-dimensional array whose rows ``arr[a]`` are contiguous but
    lie further apart than their length (``stored[:, begin:end]``)."""
    if arr.ndim != 3 or arr.flags.c_contiguous:
        return False
    if not any_size and arr.nbytes < _ROWS_UPLOAD_MIN_BYTES:
        return False
    item = arr.itemsize
    return (arr.strides[2] == item and arr.strides[1] == arr.shape[2] * item
            and arr.strides[0] > arr.shape[1] * arr.strides[1] and arr.shape[0] > 0)


class Event:
    """HIP event on the library stream."""

    def __init__(self):
        ptr = ctypes.c_void_p()
        check(_lib.load().mdx_event_create(ctypes.byref(ptr)))
        self.ptr = ptr.value

    def record(self) -> None:
        check(_lib.load().mdx_event_record(self.ptr))

    def elapsed_ms(self, stop: "Event") -> float:
        ms = ctypes.c_float(0.0)
        check(_lib.load().mdx_event_elapsed_ms(self.ptr, stop.ptr, ctypes.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _lib.load().mdx_event_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


# ----------------------------------------------------------------------- RDF


class RdfEngine:
    """Pair-distance histograms for one set of species on the current device."""

    def __init__(self, n_per_species: Sequence[int], box, cutoff, nbins: int,
                 strict_reference_masks: bool = True, literal_kernel: bool = False,
                 table_kernel: bool = False):
        self.n_per_species = [int(n) for n in n_per_species]
        self.n_species = len(self.n_per_species)
        self.n_keys = self.n_species * (self.n_species + 1) // 2
        self.nbins = int(nbins)
        flags = 0
        if strict_reference_masks:
            flags |= MDX_RDF_STRICT_REFERENCE_MASKS
        if literal_kernel:
            flags |= MDX_RDF_LITERAL_KERNEL
        if table_kernel:
            flags |= MDX_RDF_TABLE_KERNEL
        nps = (ctypes.c_int64 * self.n_species)(*self.n_per_species)
        box32 = np.asarray(box, dtype=np.float32)
        if box32.shape != (3,):
            raise ValueError("box must have three entries")
        boxc = (ctypes.c_float * 3)(*[float(b) for b in box32])
        handle = ctypes.c_void_p()
        check(_lib.load().mdx_rdf_create(ctypes.byref(handle), self.n_species, nps, boxc,
                                         ctypes.c_float(float(np.float32(cutoff))),
                                         self.nbins, flags))
        self._handle = handle.value

    @property
    def num_tiles(self) -> int:
        n = ctypes.c_int64(0)
        check(_lib.load().mdx_rdf_num_tiles(self._handle, ctypes.byref(n)))
        return int(n.value)

    def accumulate(self, species_positions: Sequence[ArrayLike], frames=None,
                   shard_rank: int = 0, shard_count: int = 1, frame_range=None) -> None:
        """Add the histogram of the selected frames.

        ``species_positions[s]`` is ``[n_s, F, 3]`` (all on the host or all on the
        device, one dtype); ``frames`` indexes axis 1 (default: every frame);
        ``frame_range = (begin, end)`` selects consecutive frames instead — host arrays
        (memory maps included) are then streamed block by block, without a copy of the
        range and without the whole array going to the device.
        """
        if len(species_positions) != self.n_species:
            raise ValueError("one position array per species is required")
        infos = [_as_input(p) for p in species_positions]
        locations = {i[2] for i in infos}
        dtypes = {i[1] for i in infos}
        if len(locations) != 1 or len(dtypes) != 1:
            raise ValueError("all species arrays must share location and dtype")
        strides = set()
        for info, n in zip(infos, self.n_per_species):
            shape = info[3]
            if len(shape) != 3 or shape[2] != 3 or shape[0] != n:
                raise ValueError(f"expected shape ({n}, F, 3), got {shape}")
            strides.add(shape[1])
        if len(strides) != 1:
            raise ValueError("all species arrays must hold the same number of frames")
        frame_stride = strides.pop()
        ptrs = (ctypes.c_void_p * self.n_species)(*[i[0] for i in infos])
        if frame_range is not None:
            if frames is not None:
                raise ValueError("give either frames or frame_range")
            begin, end = int(frame_range[0]), int(frame_range[1])
            if not 0 <= begin <= end <= frame_stride:
                raise ValueError(f"frame_range {frame_range} outside [0, {frame_stride}]")
            itemsize = 4 if next(iter(dtypes)) == _dtype_code(np.float32) else 8
            ptrs = (ctypes.c_void_p * self.n_species)(
                *[(i[0] + begin * 3 * itemsize) if i[0] else i[0] for i in infos])
            frames_ptr = None
            n_frames = end - begin
        elif frames is None:
            frames_ptr = None
            n_frames = frame_stride
        else:
            frames_arr = np.ascontiguousarray(frames, dtype=np.int64)
            frames_ptr = frames_arr.ctypes.data_as(_lib.c_i64_p)
            n_frames = len(frames_arr)
        check(_lib.load().mdx_rdf_accumulate(
            self._handle, ptrs, dtypes.pop(), locations.pop(), frame_stride, frames_ptr,
            n_frames, int(shard_rank), int(shard_count)))

    def read(self) -> np.ndarray:
        hist = np.zeros((self.n_keys, self.nbins), dtype=np.uint64)
        check(_lib.load().mdx_rdf_read(self._handle, hist.ctypes.data_as(_lib.c_u64_p)))
        return hist

    def reduce(self, comm) -> None:
        """Sum the histograms of all ranks in place on the device (``mdx_rdf_reduce``);
        ``comm`` is the ``mdx_comm_t`` handle of ``parallel.communicator()``."""
        check(_lib.load().mdx_rdf_reduce(self._handle, comm))

    def reset(self) -> None:
        check(_lib.load().mdx_rdf_reset(self._handle))

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _lib.load().mdx_rdf_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- reductions


def _out_buffer(out: Optional[np.ndarray], shape, accumulate: bool) -> np.ndarray:
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs an `out` array")
        return np.zeros(shape, dtype=np.float64)
    if out.dtype != np.float64 or not out.flags.c_contiguous or out.shape != tuple(shape):
        raise ValueError(f"out must be a C-contiguous float64 array of shape {shape}")
    return out


def weighted_atom_sum(x: ArrayLike, scalar_weight: float = 1.0,
                      weights: Optional[np.ndarray] = None, out=None,
                      accumulate: bool = False):
    """``out[t, c] (+)= sum_a w_a * x[a, t, c]`` in float64; x is ``[N, T, k]``.

    ``out`` may be a float64 ``DeviceArray`` of shape ``(T, k)``: the series then stays
    in device memory (for a following ``map_coordinates`` / ``acf_windows``)."""
    ptr, dt, loc, shape, keep = _as_input(x, rows=True)
    if len(shape) != 3:
        raise ValueError("x must have shape (n_atoms, n_frames, k)")
    n_atoms, n_frames, k = shape
    if isinstance(out, DeviceArray):
        if out.dtype != np.float64 or out.shape != (n_frames, k):
            raise ValueError(f"out must be a float64 device array of shape {(n_frames, k)}")
        out_ptr, out_loc = out.ptr, MDX_DEVICE
    else:
        out = _out_buffer(out, (n_frames, k), accumulate)
        out_ptr, out_loc = out.ctypes.data, MDX_HOST
    wptr = None
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
        if len(w) != n_atoms:
            raise ValueError("weights must have one entry per atom")
        wptr = w.ctypes.data_as(_lib.c_f64_p)
    check(_lib.load().mdx_weighted_atom_sum(ptr, dt, loc, n_atoms, n_frames, k, wptr,
                                            float(scalar_weight), out_ptr, out_loc,
                                            1 if accumulate else 0))
    return out


def momentum_flux(stress: ArrayLike, out: Optional[np.ndarray] = None,
                  accumulate: bool = False) -> np.ndarray:
    ptr, dt, loc, shape, keep = _as_input(stress, rows=True)
    if len(shape) != 3 or shape[2] != 6:
        raise ValueError("stress must have shape (n_atoms, n_frames, 6)")
    out = _out_buffer(out, (shape[1], 3), accumulate)
    check(_lib.load().mdx_momentum_flux(ptr, dt, loc, shape[0], shape[1], out.ctypes.data,
                                        MDX_HOST, 1 if accumulate else 0))
    return out


def thermal_flux(stress: ArrayLike, vel: ArrayLike, ke: ArrayLike, pe: ArrayLike,
                 out: Optional[np.ndarray] = None, accumulate: bool = False) -> np.ndarray:
    infos = _as_inputs((stress, vel, ke, pe))
    if len({i[1] for i in infos}) != 1 or len({i[2] for i in infos}) != 1:
        raise ValueError("all thermal-flux inputs must share dtype and location")
    n_atoms, n_frames = infos[0][3][0], infos[0][3][1]
    expect = [(n_atoms, n_frames, 6), (n_atoms, n_frames, 3), (n_atoms, n_frames, 1),
              (n_atoms, n_frames, 1)]
    for info, shp in zip(infos, expect):
        if tuple(info[3]) != shp:
            raise ValueError(f"expected shape {shp}, got {info[3]}")
    out = _out_buffer(out, (n_frames, 3), accumulate)
    check(_lib.load().mdx_thermal_flux(infos[0][0], infos[1][0], infos[2][0], infos[3][0],
                                       infos[0][1], infos[0][2], n_atoms, n_frames,
                                       out.ctypes.data, MDX_HOST, 1 if accumulate else 0))
    return out


def energy_weighted_sum(x: ArrayLike, ke: ArrayLike, pe: ArrayLike,
                        out: Optional[np.ndarray] = None, accumulate: bool = False):
    """``out[t, :] (+)= sum_a x[a, t, :] * (ke[a, t] + pe[a, t])`` in float64."""
    infos = _as_inputs((x, ke, pe))
    if len({i[1] for i in infos}) != 1 or len({i[2] for i in infos}) != 1:
        raise ValueError("all inputs must share dtype and location")
    n_atoms, n_frames = infos[0][3][0], infos[0][3][1]
    for info, shp in zip(infos, [(n_atoms, n_frames, 3), (n_atoms, n_frames, 1),
                                 (n_atoms, n_frames, 1)]):
        if tuple(info[3]) != shp:
            raise ValueError(f"expected shape {shp}, got {info[3]}")
    out = _out_buffer(out, (n_frames, 3), accumulate)
    check(_lib.load().mdx_energy_weighted_sum(infos[0][0], infos[1][0], infos[2][0],
                                              infos[0][1], infos[0][2], n_atoms, n_frames,
                                              out.ctypes.data, MDX_HOST,
                                              1 if accumulate else 0))
    return out


def kinaci_heat_current(pos: ArrayLike, vel: ArrayLike, force: ArrayLike, pe: ArrayLike,
                        dt: float, carry: Optional[np.ndarray] = None,
                        out: Optional[np.ndarray] = None, accumulate: bool = False):
    """``out[t, :] (+)= sum_a pos[a, t, :] * (I_a(t) + pe[a, t])`` with the running
    integral ``I_a(t) = carry[a] + dt * cumsum_t(force[a] . vel[a])``; see
    ``mdx_kinaci_heat_current``.  Returns ``(out, carry at the last frame)``."""
    infos = _as_inputs((pos, vel, force, pe))
    if len({i[1] for i in infos}) != 1 or len({i[2] for i in infos}) != 1:
        raise ValueError("all inputs must share dtype and location")
    n_atoms, n_frames = infos[0][3][0], infos[0][3][1]
    for info, shp in zip(infos, [(n_atoms, n_frames, 3)] * 3 + [(n_atoms, n_frames, 1)]):
        if tuple(info[3]) != shp:
            raise ValueError(f"expected shape {shp}, got {info[3]}")
    have_carry = carry is not None
    if have_carry:
        carry = np.array(carry, dtype=np.float64).reshape(-1)
        if carry.shape[0] != n_atoms:
            raise ValueError(f"carry must hold {n_atoms} values, got {carry.shape[0]}")
    else:
        carry = np.zeros(n_atoms, dtype=np.float64)
    out = _out_buffer(out, (n_frames, 3), accumulate)
    check(_lib.load().mdx_kinaci_heat_current(
        infos[0][0], infos[1][0], infos[2][0], infos[3][0], infos[0][1], infos[0][2], n_atoms,
        n_frames, float(dt), 1 if have_carry else 0, carry.ctypes.data, out.ctypes.data,
        MDX_HOST, 1 if accumulate else 0))
    return out, carry


# ------------------------------------------------ angular distribution function

def adf_accumulate(species_positions: Sequence[ArrayLike], box, cutoff: float,
                   bin_edges: np.ndarray, norm_power: float = 4.0, frames=None):
    """Weighted angle histograms of one batch of frames; see ``mdx_adf_accumulate``.

    ``species_positions[s]`` is ``[n_s, F, 3]``.  Returns ``(hist, triples)``: float64
    ``[n_combinations, nbins]`` weight sums and uint64 ``[n_combinations]`` triple counts,
    combinations with replacement of the species in their given order.
    """
    infos = [_as_input(p) for p in species_positions]
    if len({i[1] for i in infos}) != 1 or len({i[2] for i in infos}) != 1:
        raise ValueError("all species arrays must share location and dtype")
    n_species = len(infos)
    strides = set()
    for info in infos:
        if len(info[3]) != 3 or info[3][2] != 3:
            raise ValueError(f"expected shape (n, F, 3), got {info[3]}")
        strides.add(info[3][1])
    if len(strides) != 1:
        raise ValueError("all species arrays must hold the same number of frames")
    frame_stride = strides.pop()
    n_per_species = np.ascontiguousarray([i[3][0] for i in infos], dtype=np.int64)
    ptrs = (ctypes.c_void_p * n_species)(*[i[0] for i in infos])
    if frames is None:
        frames_ptr, n_frames = None, frame_stride
    else:
        frames_arr = np.ascontiguousarray(frames, dtype=np.int64)
        frames_ptr, n_frames = frames_arr.ctypes.data_as(_lib.c_i64_p), len(frames_arr)
    edges = np.ascontiguousarray(bin_edges, dtype=np.float64)
    nbins = len(edges) - 1
    lib = _lib.load()
    n_combos = lib.mdx_adf_num_combinations(n_species)
    hist = np.zeros((n_combos, nbins), dtype=np.float64)
    triples = np.zeros(n_combos, dtype=np.uint64)
    box_arr = np.ascontiguousarray(np.asarray(box, dtype=np.float32).reshape(-1))
    if box_arr.size != 3:
        raise ValueError("box must have three entries")
    check(lib.mdx_adf_accumulate(
        ptrs, infos[0][1], infos[0][2], n_species, n_per_species.ctypes.data_as(_lib.c_i64_p),
        frame_stride, frames_ptr, n_frames,
        box_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), float(np.float32(cutoff)),
        nbins, edges.ctypes.data_as(_lib.c_f64_p), float(norm_power),
        hist.ctypes.data_as(_lib.c_f64_p), triples.ctypes.data_as(_lib.c_u64_p)))
    return hist, triples


# ------------------------------------------------- coordinate transformations

MAP_WRAP, MAP_WRAP_CENTERED, MAP_SCALE, MAP_UNWRAP_INDICES, MAP_VELOCITY = range(5)


def _map_output(out, shape, out_dtype):
    """(pointer, dtype code, location, array to return) of a transformation output."""
    if isinstance(out, DeviceArray):
        if tuple(out.shape) != tuple(shape):
            raise ValueError(f"out must have shape {tuple(shape)}")
        return out.ptr, _dtype_code(out.dtype), MDX_DEVICE, out
    if out is None:
        out = np.empty(shape, dtype=out_dtype)
    if out.dtype not in (np.float32, np.float64) or not out.flags.c_contiguous \
            or out.shape != tuple(shape):
        raise ValueError(f"out must be a C-contiguous float array of shape {tuple(shape)}")
    return out.ctypes.data, _dtype_code(out.dtype), MDX_HOST, out


def _three(values, name: str) -> np.ndarray:
    arr = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
    if arr.size != 3:
        raise ValueError(f"{name} must have three entries")
    return arr


def unwrap_coordinates(pos: ArrayLike, box, carry: Optional[dict] = None, out=None,
                       out_dtype=np.float64):
    """Unwrap ``pos`` ``[N, T, 3]`` by jump detection; returns ``(unwrapped, carry)``
    with ``carry = {"last_pos": [N, 3], "last_image_box": [N, 3]}`` for the next batch."""
    ptr, dt, loc, shape, keep = _as_input(pos, rows=True)
    if len(shape) != 3 or shape[2] != 3:
        raise ValueError("pos must have shape (n_atoms, n_frames, 3)")
    n_atoms, n_frames = shape[0], shape[1]
    box_arr = _three(box, "box")
    if carry is None:
        last_pos = np.zeros((n_atoms, 3))
        last_image = np.zeros((n_atoms, 3))
    else:
        last_pos = np.array(carry["last_pos"], dtype=np.float64).reshape(n_atoms, 3)
        last_image = np.array(carry["last_image_box"], dtype=np.float64).reshape(n_atoms, 3)
    optr, odt, oloc, result = _map_output(out, shape, out_dtype)
    check(_lib.load().mdx_unwrap_coordinates(
        ptr, dt, loc, n_atoms, n_frames, box_arr.ctypes.data_as(_lib.c_f64_p),
        0 if carry is None else 1, last_pos.ctypes.data_as(_lib.c_f64_p),
        last_image.ctypes.data_as(_lib.c_f64_p), optr, odt, oloc))
    return result, {"last_pos": last_pos, "last_image_box": last_image}


def map_coordinates(op: int, x: ArrayLike, params, aux: Optional[ArrayLike] = None, out=None,
                    out_dtype=np.float64):
    """Element-wise coordinate maps (``MAP_*``) of ``x`` ``[N, T, 3]``."""
    ptr, dt, loc, shape, keep = _as_input(x)
    if len(shape) != 3 or shape[2] != 3:
        raise ValueError("x must have shape (n_atoms, n_frames, 3)")
    aptr, keep_aux = None, None
    if aux is not None:
        if isinstance(x, DeviceArray) != isinstance(aux, DeviceArray):
            raise ValueError("aux must live where x lives")
        if not isinstance(aux, DeviceArray):
            aux = np.asarray(aux, dtype=keep.dtype)
        aptr, adt, aloc, ashape, keep_aux = _as_input(aux)
        if tuple(ashape) != tuple(shape) or adt != dt:
            raise ValueError("aux must have the shape and dtype of x")
    par = _three(params, "params")
    optr, odt, oloc, result = _map_output(out, shape, out_dtype)
    check(_lib.load().mdx_map_coordinates(int(op), ptr, aptr, dt, loc, shape[0], shape[1],
                                          par.ctypes.data_as(_lib.c_f64_p), optr, odt, oloc))
    return result


# -------------------------------------------------------------- correlations


@dataclasses.dataclass
class Moments:
    """Streaming mean / sum of squared deviations over ``n`` windows."""

    n: int
    mean: np.ndarray
    m2: np.ndarray

    def merge(self, other: "Moments") -> "Moments":
        """Parallel-variance (Chan et al.) combination of two window sets."""
        if other.n == 0:
            return self
        if self.n == 0:
            return other
        n = self.n + other.n
        delta = other.mean - self.mean
        mean = self.mean + delta * (other.n / n)
        m2 = self.m2 + other.m2 + delta * delta * (self.n * other.n / n)
        return Moments(n, mean, m2)

    def std(self) -> np.ndarray:
        """Population standard deviation, ``np.std(..., axis=0)`` of the windows."""
        return np.sqrt(self.m2 / self.n)


def _tau_array(tau) -> np.ndarray:
    tau = np.ascontiguousarray(tau, dtype=np.int32)
    if tau.ndim != 1 or len(tau) == 0:
        raise ValueError("tau must be a non-empty 1-D index array")
    return tau


def acf_windows(x: ArrayLike, correlation_time: int, win_begin: int, win_end: int, tau,
                scale: float = 1.0, integral_scale: float = 1.0,
                time: Optional[np.ndarray] = None):
    """Windowed autocorrelation of ``x`` ``[S, T, k]``; see ``mdx_acf_windows``.

    Returns ``(acf_sum, moments)``; ``moments`` is ``None`` without ``time``.
    """
    ptr, dt, loc, shape, keep = _as_input(x, rows=True)
    if len(shape) != 3:
        raise ValueError("x must have shape (n_series, n_frames, k)")
    tau = _tau_array(tau)
    R = len(tau)
    acf = np.zeros(R, dtype=np.float64)
    n_win = ctypes.c_int64(0)
    if time is not None:
        time_arr = np.ascontiguousarray(time, dtype=np.float64)
        if time_arr.shape != (R,):
            raise ValueError("time must have one entry per tau value")
        mean = np.zeros(max(R - 1, 1), dtype=np.float64)
        m2 = np.zeros(max(R - 1, 1), dtype=np.float64)
        tptr = time_arr.ctypes.data_as(_lib.c_f64_p)
        mptr = mean.ctypes.data_as(_lib.c_f64_p)
        m2ptr = m2.ctypes.data_as(_lib.c_f64_p)
    else:
        tptr = mptr = m2ptr = None
    check(_lib.load().mdx_acf_windows(
        ptr, dt, loc, shape[0], shape[1], shape[2], int(correlation_time), int(win_begin),
        int(win_end), tau.ctypes.data_as(_lib.c_i32_p), R, float(scale),
        float(integral_scale), tptr, acf.ctypes.data_as(_lib.c_f64_p), mptr, m2ptr,
        ctypes.byref(n_win)))
    moments = None
    if time is not None:
        moments = Moments(int(n_win.value), mean[:R - 1], m2[:R - 1])
    return acf, moments


def ccf_windows(a: ArrayLike, b: ArrayLike, correlation_time: int, win_begin: int,
                win_end: int, data_range: int, cross_weight: float = 1.0,
                self_atoms: Optional[ArrayLike] = None, self_weight: float = 0.0,
                time: Optional[np.ndarray] = None):
    """Windowed cross-correlation of the series ``a``, ``b`` ``[T, k]`` minus
    ``self_weight`` times the summed autocorrelation of ``self_atoms`` ``[N, T, k]``,
    without lag normalisation; see ``mdx_ccf_windows``.  Returns ``(cf_sum, moments)``."""
    infos = [_as_input(a)]
    infos.append(infos[0] if b is a else _as_input(b))
    if self_atoms is not None:
        infos.append(_as_input(self_atoms))
    if len({i[1] for i in infos}) != 1 or len({i[2] for i in infos}) != 1:
        raise ValueError("all inputs must share dtype and location")
    shape = tuple(infos[0][3])
    if len(shape) != 2 or tuple(infos[1][3]) != shape:
        raise ValueError("a and b must both have shape (n_frames, k)")
    n_self, sptr = 0, None
    if self_atoms is not None:
        sshape = tuple(infos[2][3])
        if len(sshape) != 3 or sshape[1:] != shape:
            raise ValueError("self_atoms must have shape (n_atoms, n_frames, k)")
        n_self, sptr = sshape[0], infos[2][0]
    R = int(data_range)
    cf = np.zeros(R, dtype=np.float64)
    n_win = ctypes.c_int64(0)
    if time is not None:
        time_arr = np.ascontiguousarray(time, dtype=np.float64)
        if time_arr.shape != (R,):
            raise ValueError("time must have one entry per lag")
        mean = np.zeros(max(R - 1, 1), dtype=np.float64)
        m2 = np.zeros(max(R - 1, 1), dtype=np.float64)
        tptr = time_arr.ctypes.data_as(_lib.c_f64_p)
        mptr = mean.ctypes.data_as(_lib.c_f64_p)
        m2ptr = m2.ctypes.data_as(_lib.c_f64_p)
    else:
        tptr = mptr = m2ptr = None
    check(_lib.load().mdx_ccf_windows(
        infos[0][0], infos[1][0], sptr, infos[0][1], infos[0][2], n_self, shape[0], shape[1],
        int(correlation_time), int(win_begin), int(win_end), R, float(cross_weight),
        float(self_weight), tptr, cf.ctypes.data_as(_lib.c_f64_p), mptr, m2ptr,
        ctypes.byref(n_win)))
    moments = None
    if time is not None:
        moments = Moments(int(n_win.value), mean[:R - 1], m2[:R - 1])
    return cf, moments


def msd_windows(x: ArrayLike, correlation_time: int, win_begin: int, win_end: int, tau,
                scale: float = 1.0) -> np.ndarray:
    """Windowed squared displacement sums of ``x`` ``[S, T, k]``; see ``mdx_msd_windows``."""
    ptr, dt, loc, shape, keep = _as_input(x, rows=True)
    if len(shape) != 3:
        raise ValueError("x must have shape (n_series, n_frames, k)")
    tau = _tau_array(tau)
    msd = np.zeros(len(tau), dtype=np.float64)
    check(_lib.load().mdx_msd_windows(
        ptr, dt, loc, shape[0], shape[1], shape[2], int(correlation_time), int(win_begin),
        int(win_end), tau.ctypes.data_as(_lib.c_i32_p), len(tau), float(scale),
        msd.ctypes.data_as(_lib.c_f64_p)))
    return msd


# ------------------------------------------------------------------ profiling

SLOTS = {"rdf_pack": 0, "rdf_tile": 1, "atom_sum": 2, "acf_partial": 3, "acf_finish": 4,
         "msd": 5, "flux": 6, "transform": 7}


def profile_enable(on: bool = True) -> None:
    """Bracket every kernel launch with HIP events on the library stream."""
    check(_lib.load().mdx_profile_enable(1 if on else 0))


def profile_reset() -> None:
    check(_lib.load().mdx_profile_reset())


def profile_query(slot: str):
    """(total kernel milliseconds, launches) of one slot since the last reset."""
    ms = ctypes.c_double(0.0)
    n = ctypes.c_int64(0)
    check(_lib.load().mdx_profile_query(SLOTS[slot], ctypes.byref(ms), ctypes.byref(n)))
    return float(ms.value), int(n.value)
