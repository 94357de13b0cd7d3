import time, numpy as np
import lammps_analysis_amd.engine as engine
N, F, D = 5000, 16384, 500
vel = engine.DeviceArray((N, F, 3), np.float32); vel.fill_uniform(seed=31, lo=-1.0, hi=1.0)
tau = np.arange(D); tm = tau * 0.002; n_win = F - D
def t(fn, n=5):
    fn(); engine.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    engine.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
ms, cur = t(lambda: engine.weighted_atom_sum(vel, scalar_weight=1.0)); print("sum", ms)
ms, cur2 = t(lambda: engine.weighted_atom_sum(vel, scalar_weight=-1.0, out=cur, accumulate=True)); print("sum acc", ms)
ms, series = t(lambda: cur.astype(np.float32).astype(np.float64)[None]); print("astype", ms)
ms, _ = t(lambda: engine.acf_windows(series, 1, 0, n_win, tau, time=tm)); print("acf", ms)
ms, _ = t(lambda: engine.acf_windows(series, 1, 0, n_win, tau)); print("acf no integral", ms)
