import time, numpy as np
import lammps_analysis_amd.engine as engine
x = np.random.default_rng(0).random((5000, 16384, 3), dtype=np.float32)   # 983 MB
for _ in range(3):
    t0 = time.perf_counter(); d = engine.DeviceArray.from_numpy(x); engine.synchronize(); dt = time.perf_counter() - t0
    print(f"from_numpy {x.nbytes/dt/1e9:.1f} GB/s"); 
    back = d.to_numpy(); assert np.array_equal(back, x); d.free()
odd = np.random.default_rng(1).random(100_000_037 // 4 * 4 + 3).astype(np.float32)[:-1]  # not a multiple of the chunk
d = engine.DeviceArray.from_numpy(odd); assert np.array_equal(d.to_numpy(), odd); d.free()
for _ in range(3):
    t0 = time.perf_counter(); s = engine.weighted_atom_sum(x); dt = time.perf_counter() - t0
    print(f"weighted_atom_sum from host: {dt*1e3:.1f} ms -> {x.nbytes/dt/1e9:.1f} GB/s")
assert np.allclose(s, x.astype(np.float64).sum(axis=0), rtol=1e-12)
tau = np.arange(500)
for _ in range(2):
    t0 = time.perf_counter(); engine.acf_windows(x, 1, 0, 16384 - 500, tau); dt = time.perf_counter() - t0
    print(f"per-atom acf from host: {dt*1e3:.1f} ms")
