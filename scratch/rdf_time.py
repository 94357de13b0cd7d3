"""Times the RDF kernels on the configs[1] geometry: python scratch/rdf_time.py [frames]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lammps_analysis_amd import engine
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
engine.set_device(0)
na = engine.DeviceArray((5000, nf, 3), np.float32); na.fill_uniform(1, 0.0, 65.0)
cl = engine.DeviceArray((5000, nf, 3), np.float32); cl.fill_uniform(2, 0.0, 65.0)
ref = None
for name, kw in (("fast", {}), ("table", {"table_kernel": True})):
    rdf = engine.RdfEngine([5000, 5000], [65.0] * 3, np.float32(32.4), 3240, **kw)
    rdf.accumulate([na, cl]); engine.synchronize(); rdf.reset()
    engine.profile_enable(True); engine.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        rdf.accumulate([na, cl])
    engine.synchronize()
    dt = (time.perf_counter() - t0) / 3
    ms, n = engine.profile_query("rdf_tile")
    h = rdf.read()
    if ref is None:
        ref = h
    print(f"{name}: {nf / dt:.0f} frames/s, kernel {ms / n:.2f} ms per launch, equal={np.array_equal(h, ref)}, counts={h.sum()}")
    rdf.close()
