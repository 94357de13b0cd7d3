"""Run a timing script against an alternative build: ALT=path/to/libmdx.so python scratch/ab.py script.py"""
import os, pathlib, runpy, sys
import lammps_analysis_amd._lib as L
if os.environ.get("ALT"):
    L.LIB_PATH = pathlib.Path(os.environ["ALT"]).resolve()
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
