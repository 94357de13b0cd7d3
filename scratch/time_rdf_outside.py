"""RDF step rate on frames whose atoms lie up to half an Angstrom outside the box (spread
of more than one box length: the frame class between "compact" and "boxes away").
usage: python scratch/time_rdf_outside.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lammps_analysis_amd.engine as engine

F = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
box, cutoff = 65.0, np.float32(32.4)
for lo, hi, what in ((0.0, 65.0, "inside the box"), (-0.5, 65.5, "up to 0.5 outside"),
                     (-15.0, 80.0, "up to 15 outside")):
    species = []
    for s in range(2):
        a = engine.DeviceArray((5000, F, 3), np.float32)
        a.fill_uniform(seed=50 + s, lo=lo, hi=hi)
        species.append(a)
    eng = engine.RdfEngine([5000, 5000], [box] * 3, cutoff, 3240, strict_reference_masks=True)
    eng.accumulate(species)
    engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.accumulate(species)
    engine.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{what}: {F / dt:.0f} frames/s", flush=True)
    eng.close()
    for a in species:
        a.free()
