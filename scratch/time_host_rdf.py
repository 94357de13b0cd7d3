import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np
import lammps_analysis_amd.engine as engine
N, F = 5000, 1024
rng = np.random.default_rng(0)
host = [(rng.random((N, F, 3), dtype=np.float32) * 65.0) for _ in range(2)]
dev = [engine.DeviceArray.from_numpy(h) for h in host]
eng = engine.RdfEngine([N, N], [65.0]*3, np.float32(32.4), 3240)
def t(fn, n=3):
    fn(); engine.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    engine.synchronize(); return (time.perf_counter() - t0) / n
td = t(lambda: eng.accumulate(dev)); th = t(lambda: eng.accumulate(host))
print(f"device-resident {F/td:.0f} frames/s, host input {F/th:.0f} frames/s ({th*1e3:.1f} ms vs {td*1e3:.1f} ms)")
