"""Build libmdx.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.

The library cross-compiles on a machine without a GPU.  ``-ffp-contract=off`` is
part of the numerical contract: the RDF kernel must round every float32 multiply
before the following add/subtract, exactly like the reference's separate
TensorFlow ops.
"""
from __future__ import annotations

import concurrent.futures
import os
import pathlib
import shutil
import subprocess

PACKAGE_DIR = pathlib.Path(__file__).resolve().parent
REPO_ROOT = PACKAGE_DIR.parent
CSRC = PACKAGE_DIR / "csrc"
LIB_DIR = PACKAGE_DIR / "lib"
LIB_PATH = LIB_DIR / "libmdx.so"
OBJ_DIR = PACKAGE_DIR / "build"

SOURCES = ["mdx_runtime.hip", "mdx_rdf.hip", "mdx_reduce.hip", "mdx_corr.hip",
           "mdx_transform.hip", "mdx_adf.hip", "mdx_comm.hip"]

HIPCC_FLAGS = [
    "-O3",
    "-std=c++17",
    "--offload-arch=gfx950",
    "-fPIC",
    "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libmdx.so cannot be built")
    return exe


def _newest_source_mtime() -> float:
    paths = [CSRC / s for s in SOURCES] + list(CSRC.glob("*.h"))
    paths.append(REPO_ROOT / "include" / "mdx.h")
    return max(p.stat().st_mtime for p in paths)


def is_up_to_date() -> bool:
    return LIB_PATH.exists() and LIB_PATH.stat().st_mtime >= _newest_source_mtime()


def build_library(force: bool = False, verbose: bool = False) -> pathlib.Path:
    """Compile every translation unit and link ``lib/libmdx.so``."""
    if not force and is_up_to_date():
        return LIB_PATH
    hipcc = _hipcc()
    LIB_DIR.mkdir(exist_ok=True)
    OBJ_DIR.mkdir(exist_ok=True)
    include = ["-I", str(REPO_ROOT / "include"), "-I", str(CSRC)]

    def compile_one(src: str) -> pathlib.Path:
        obj = OBJ_DIR / (src + ".o")
        cmd = [hipcc, *HIPCC_FLAGS, *include, "-c", str(CSRC / src), "-o", str(obj)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
        if verbose and res.stderr.strip():
            print(res.stderr)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        objects = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH)]
    cmd += [str(o) for o in objects] + ["-ldl"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"linking libmdx.so failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
