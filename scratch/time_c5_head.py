import os, time, sys, numpy as np
import lammps_analysis_amd.engine as engine
T = 1_000_000
x = np.random.default_rng(5).standard_normal((1, T, 3)).astype(np.float32)
dev = engine.DeviceArray.from_numpy(x)
for D in (500, 5000):
    tau = np.arange(D); tm = tau * 0.002
    for n_win in (8192, 16384, 65536, 262144, T - D):
        for use_time in (False, True):
            def run(): return engine.acf_windows(dev, 1, 0, n_win, tau, time=tm if use_time else None)
            run(); engine.synchronize()
            t0 = time.perf_counter(); n = 3
            for _ in range(n): run()
            engine.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
            print(f"D={D} n_win={n_win} integral={use_time} {ms:.2f} ms", flush=True)
