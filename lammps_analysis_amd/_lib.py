"""ctypes binding of libmdx.so, the HIP kernel library (see include/mdx.h).

There is no CPU fallback: if the shared library is missing or no GPU is usable,
every compute entry point raises.  The library is built in-tree by
``lammps_analysis_amd.build.build_library()``.
"""
from __future__ import annotations

import ctypes
import pathlib
import threading

LIB_PATH = pathlib.Path(__file__).resolve().parent / "lib" / "libmdx.so"

MDX_OK = 0
MDX_ERR_INVALID_ARGUMENT = -1
MDX_ERR_DEVICE = -2
MDX_ERR_OUT_OF_MEMORY = -3
MDX_ERR_STATE = -4

MDX_HOST = 0
MDX_DEVICE = 1
MDX_F32 = 0
MDX_F64 = 1

MDX_COMM_ID_BYTES = 128
MDX_COMM_U64 = 0
MDX_COMM_I64 = 1
MDX_COMM_F64 = 2
MDX_COMM_SUM = 0
MDX_COMM_MAX = 1

MDX_RDF_STRICT_REFERENCE_MASKS = 1
MDX_RDF_LITERAL_KERNEL = 2
MDX_RDF_TABLE_KERNEL = 4


class MdxError(RuntimeError):
    """A libmdx call failed (HIP error, out of memory, ...)."""


class MdxLibraryMissing(ImportError):
    """libmdx.so has not been built; the engine has no other compute path."""


c_void_pp = ctypes.POINTER(ctypes.c_void_p)
c_i64_p = ctypes.POINTER(ctypes.c_int64)
c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_u64_p = ctypes.POINTER(ctypes.c_uint64)
c_f32_p = ctypes.POINTER(ctypes.c_float)
c_f64_p = ctypes.POINTER(ctypes.c_double)


class RdfFastPlan(ctypes.Structure):
    """``mdx_rdf_fast_plan_t`` of include/mdx.h."""

    _fields_ = [("usable", ctypes.c_int), ("frac_bits", ctypes.c_int),
                ("edge_units", ctypes.c_uint32), ("keep_bits", ctypes.c_uint32),
                ("magic", ctypes.c_float), ("scale", ctypes.c_float),
                ("s_cut", ctypes.c_float), ("half_box", ctypes.c_float * 3),
                ("step", ctypes.c_double)]


# name -> (restype, argtypes); mirrors include/mdx.h one to one
SIGNATURES = {
    "mdx_version": (ctypes.c_int, []),
    "mdx_last_error": (ctypes.c_char_p, []),
    "mdx_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "mdx_set_device": (ctypes.c_int, [ctypes.c_int]),
    "mdx_device_info": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_int), c_u64_p]),
    "mdx_synchronize": (ctypes.c_int, []),
    "mdx_malloc": (ctypes.c_int, [c_void_pp, ctypes.c_size_t]),
    "mdx_free": (ctypes.c_int, [ctypes.c_void_p]),
    "mdx_memcpy_h2d": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "mdx_memcpy_h2d_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_size_t, ctypes.c_size_t]),
    "mdx_memcpy_d2h": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "mdx_memset": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]),
    "mdx_release_cached_memory": (ctypes.c_int, []),
    "mdx_allocation_counters": (ctypes.c_int, [c_u64_p, c_u64_p, c_u64_p]),
    "mdx_event_create": (ctypes.c_int, [c_void_pp]),
    "mdx_event_record": (ctypes.c_int, [ctypes.c_void_p]),
    "mdx_event_elapsed_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_f32_p]),
    "mdx_event_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mdx_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "mdx_profile_reset": (ctypes.c_int, []),
    "mdx_profile_query": (ctypes.c_int, [ctypes.c_int, c_f64_p, c_i64_p]),
    "mdx_fill_uniform_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64,
                                            ctypes.c_uint64, ctypes.c_float,
                                            ctypes.c_float]),
    "mdx_fill_ou_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int, ctypes.c_uint64, ctypes.c_float,
                                       ctypes.c_float]),
    "mdx_rdf_create": (ctypes.c_int, [c_void_pp, ctypes.c_int, c_i64_p, c_f32_p,
                                      ctypes.c_float, ctypes.c_int, ctypes.c_uint32]),
    "mdx_rdf_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mdx_rdf_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "mdx_rdf_num_keys": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "mdx_rdf_num_tiles": (ctypes.c_int, [ctypes.c_void_p, c_i64_p]),
    "mdx_rdf_accumulate": (ctypes.c_int, [ctypes.c_void_p, c_void_pp, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int64, c_i64_p,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "mdx_rdf_read": (ctypes.c_int, [ctypes.c_void_p, c_u64_p]),
    "mdx_rdf_fast_plan": (ctypes.c_int, [c_f32_p, ctypes.c_float, ctypes.c_int,
                                         ctypes.POINTER(RdfFastPlan)]),
    "mdx_comm_available": (ctypes.c_int, []),
    "mdx_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "mdx_comm_create": (ctypes.c_int, [c_void_pp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_size_t]),
    "mdx_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mdx_comm_rank": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_int)]),
    "mdx_comm_allreduce": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "mdx_rdf_reduce": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "mdx_weighted_atom_sum": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                             c_f64_p, ctypes.c_double, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_int]),
    "mdx_weighted_atom_sum_static": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                    ctypes.c_int64, ctypes.c_int,
                                                    ctypes.c_double, ctypes.c_void_p,
                                                    ctypes.c_int]),
    "mdx_momentum_flux": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_int]),
    "mdx_thermal_flux": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_int]),
    "mdx_energy_weighted_sum": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int]),
    "mdx_kinaci_heat_current": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_int]),
    "mdx_adf_num_combinations": (ctypes.c_int, [ctypes.c_int]),
    "mdx_adf_accumulate": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, c_i64_p, ctypes.c_int64,
                                          c_i64_p, ctypes.c_int64,
                                          ctypes.POINTER(ctypes.c_float), ctypes.c_float,
                                          ctypes.c_int, c_f64_p, ctypes.c_float, c_f64_p,
                                          ctypes.POINTER(ctypes.c_uint64)]),
    "mdx_unwrap_coordinates": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int64, ctypes.c_int64, c_f64_p,
                                              ctypes.c_int, c_f64_p, c_f64_p, ctypes.c_void_p,
                                              ctypes.c_int, ctypes.c_int]),
    "mdx_map_coordinates": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                           ctypes.c_int64, c_f64_p, ctypes.c_void_p,
                                           ctypes.c_int, ctypes.c_int]),
    "mdx_acf_windows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       c_i32_p, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_double, c_f64_p, c_f64_p, c_f64_p, c_f64_p,
                                       c_i64_p]),
    "mdx_ccf_windows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_double, c_f64_p, c_f64_p,
                                       c_f64_p, c_f64_p, c_i64_p]),
    "mdx_msd_windows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       c_i32_p, ctypes.c_int, ctypes.c_double, c_f64_p]),
}

_lock = threading.Lock()
_lib = None


def load() -> ctypes.CDLL:
    """Load libmdx.so and declare every prototype of include/mdx.h."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise MdxLibraryMissing(
                f"{LIB_PATH} not found. Build it with "
                "`python -m lammps_analysis_amd.build` (needs hipcc); "
                "this engine has no CPU fallback."
            )
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        return lib


def check(rc: int) -> None:
    """Map a libmdx status code to a Python exception."""
    if rc == MDX_OK:
        return
    msg = load().mdx_last_error().decode("utf-8", "replace")
    if rc == MDX_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    if rc == MDX_ERR_OUT_OF_MEMORY:
        raise MemoryError(msg)
    raise MdxError(f"libmdx error {rc}: {msg}")


def device_count() -> int:
    n = ctypes.c_int(0)
    rc = load().mdx_device_count(ctypes.byref(n))
    if rc != MDX_OK:
        return 0
    return int(n.value)
