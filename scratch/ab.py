"""Run a timing script against an alternative build: ALT=path/to/libmdx.so python scratch/ab.py script.py"""
import os, pathlib, runpy, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import lammps_analysis_amd._lib as L
if os.environ.get("ALT"):
    L.LIB_PATH = pathlib.Path(os.environ["ALT"]).resolve()
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
