"""Achieved HBM rates of the streaming kernels that the benchmark line does not time:
Kinaci heat current, unwrap, element-wise maps.  Device-resident float32 input.
usage: python scratch/time_streams.py [atoms] [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lammps_analysis_amd.engine as engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
cell = N * F * 4


def arr(k, seed, lo=-1.0, hi=1.0):
    a = engine.DeviceArray((N, F, k), np.float32)
    a.fill_uniform(seed=seed, lo=lo, hi=hi)
    return a


def t(fn, n=3, slot="transform"):
    """kernel time per call from the library's event timers"""
    fn(); engine.synchronize()
    engine.profile_enable(True); engine.profile_reset()
    for _ in range(n):
        fn()
    engine.synchronize()
    ms, launches = engine.profile_query(slot)
    engine.profile_enable(False)
    return ms * 1e-3 / max(launches, 1)


pos, vel, force, pe = arr(3, 1, 0.0, 65.0), arr(3, 2), arr(3, 3), arr(1, 4)
s = t(lambda: engine.kinaci_heat_current(pos, vel, force, pe, 0.002), slot="flux")
print(f"kinaci: {s*1e3:.2f} ms, {10*cell/s/1e12:.2f} TB/s of the 10 words per atom-frame "
      f"({16*cell/s/1e12:.2f} TB/s of what two passes fetch)", flush=True)
force.free(); pe.free(); vel.free()
y = engine.DeviceArray((N, F, 3), np.float32)
box = [65.0] * 3
for name, fn in (("unwrap", lambda: engine.unwrap_coordinates(pos, box, out=y)),
                 ("wrap", lambda: engine.map_coordinates(engine.MAP_WRAP, pos, box, out=y)),
                 ("scale", lambda: engine.map_coordinates(engine.MAP_SCALE, pos, [2.0] * 3, out=y)),
                 ("velocity", lambda: engine.map_coordinates(engine.MAP_VELOCITY, pos,
                                                             [0.002, 0, 0], out=y))):
    s = t(fn)
    print(f"{name}: {s*1e3:.2f} ms, {6*cell/s/1e12:.2f} TB/s (read + write float32)",
          flush=True)
