import time, numpy as np
import lammps_analysis_amd.engine as engine
def t(fn, n=3):
    fn(); engine.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    engine.synchronize(); return (time.perf_counter() - t0) / n * 1e3
N, F = 5000, 16384
pos = engine.DeviceArray((N, F, 3), np.float32); pos.fill_uniform(seed=3, lo=0.0, hi=65.0)
for D in (100, 500):
    tau = np.arange(D)
    print("msd N=5000 F=16384 D=%d ct=1: %.2f ms" % (D, t(lambda: engine.msd_windows(pos, 1, 0, F - D, tau))))
engine.profile_enable(True)
for T, D in ((1_000_000, 500), (1_000_000, 5000)):
    x = engine.DeviceArray((1, T, 3), np.float32); x.fill_uniform(seed=5, lo=-1.0, hi=1.0)
    tau = np.arange(D); tm = tau * 0.002
    engine.profile_reset()
    ms = t(lambda: engine.acf_windows(x, 1, 0, T - D, tau, time=tm))
    print("acf T=1e6 D=%d: %.2f ms wall, slide kernel %.2f ms" % (D, ms, engine.profile_query("acf_finish")[0] / engine.profile_query("acf_finish")[1]))
    x.free()
