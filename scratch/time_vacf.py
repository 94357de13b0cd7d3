import time, numpy as np
import lammps_analysis_amd.engine as engine
N, F, D = 5000, 100000, 500
vel = engine.DeviceArray((N, F, 3), np.float32); vel.fill_uniform(seed=31, lo=-1.0, hi=1.0)
tau = np.arange(D); tm = tau * 0.002
def run(): return engine.acf_windows(vel, 1, 0, F - D, tau, time=tm)
run(); engine.profile_enable(True); engine.profile_reset(); engine.synchronize()
t0 = time.perf_counter()
for _ in range(3): run()
engine.synchronize(); dt = (time.perf_counter() - t0) / 3
ms, n = engine.profile_query("acf_partial")
print(f"vacf {dt*1e3:.2f} ms per pass, band kernel {ms/n:.2f} ms = {2.0*F*D*N*3/(ms/n*1e-3)/1e12:.1f} TFLOP/s algorithmic, {3126*4*63750*2048/(ms/n*1e-3)/1e12:.1f} issued")
