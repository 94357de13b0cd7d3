"""Compile the C restatement (oracle/mdx_oracle.c) into oracle/liboracle.so.

TEST INFRASTRUCTURE ONLY.  The reference package is pure Python with no native
sources, so there is nothing to compile under oracle/_ref/.
"""
from __future__ import annotations

import pathlib
import shutil
import subprocess

HERE = pathlib.Path(__file__).resolve().parent
SRC = HERE / "mdx_oracle.c"
LIB = HERE / "liboracle.so"


def build_oracle(force: bool = False) -> pathlib.Path:
    if not force and LIB.exists() and LIB.stat().st_mtime >= SRC.stat().st_mtime:
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found; the C oracle cannot be built")
    cmd = [gcc, "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-shared",
           "-fPIC", "-o", str(LIB), str(SRC), "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"building the C oracle failed:\n{res.stdout}\n{res.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_oracle(force=True))
