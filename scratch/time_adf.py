import time, numpy as np
import lammps_analysis_amd.engine as engine
N, F = 5000, 16
dev = []
for s in range(2):
    d = engine.DeviceArray((N, F, 3), np.float32); d.fill_uniform(seed=60 + s, lo=0.0, hi=65.0); dev.append(d)
edges = np.histogram_bin_edges(np.zeros(1, np.float32), bins=500, range=(0.0, 3.15))
def t(fn, n=3):
    fn(); engine.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    engine.synchronize(); return (time.perf_counter() - t0) / n, r
for cutoff in (6.0, 10.0):
    dt, (hist, triples) = t(lambda: engine.adf_accumulate(dev, [65.0]*3, cutoff, edges))
    print(f"cutoff {cutoff}: {dt*1e3:.1f} ms for {F} frames of 10k atoms -> {F/dt:.0f} frames/s, {triples.sum()/F:.3g} triples/frame")
