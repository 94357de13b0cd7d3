"""Wall time of one small RDF call at the engine level: create, 500 frames of 10k atoms
from device memory, read, close.  usage: python scratch/time_rdf_small_call.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lammps_analysis_amd.engine as engine

species = []
for s in range(2):
    a = engine.DeviceArray((5000, 500, 3), np.float32)
    a.fill_uniform(seed=70 + s, lo=0.0, hi=65.0)
    species.append(a)
engine.synchronize()
for turn in range(3):
    t0 = time.perf_counter()
    eng = engine.RdfEngine([5000, 5000], [65.0] * 3, np.float32(32.4), 3240)
    t1 = time.perf_counter()
    eng.accumulate(species)
    hist = eng.read()
    t2 = time.perf_counter()
    eng.close()
    t3 = time.perf_counter()
    print(f"turn {turn}: create {1e3*(t1-t0):.1f} ms, accumulate + read {1e3*(t2-t1):.1f} ms, "
          f"close {1e3*(t3-t2):.1f} ms, pairs {int(hist.sum())}", flush=True)
