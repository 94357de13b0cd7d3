"""C5: [1, 1e6, 3] AR(1) series, data_range 500 / 2000 / 5000 / 8000, correlation_time 1."""
import os, time, sys, numpy as np
import lammps_analysis_amd.engine as engine
T = 1_000_000
rng = np.random.default_rng(5)
phi = np.exp(-1 / 200)
xi = rng.standard_normal((T, 3))
x = np.empty((T, 3))
acc = np.zeros(3)
from scipy.signal import lfilter
x = lfilter([1.0], [1.0, -phi], xi, axis=0).astype(np.float32)[None]
if os.environ.get("C5_F64") == "1": x = x.astype(np.float64)
dev = engine.DeviceArray.from_numpy(x) if hasattr(engine.DeviceArray, "from_numpy") else None
import os
NOTIME = os.environ.get("C5_NOTIME") == "1"
for D in [int(a) for a in (sys.argv[1:] or ["500", "2000", "5000", "8000"])]:
    tau = np.arange(D); tm = tau * 0.002; n_win = T - D
    src = dev if dev is not None else x
    def run(): return engine.acf_windows(src, 1, 0, n_win, tau, time=None if NOTIME else tm)
    run(); engine.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): r = run()
    engine.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"D={D} {ms:.2f} ms  {T/ms*1e3:.3e} frames/s", flush=True)
