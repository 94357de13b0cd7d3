"""Unwrap kernel rate on a wrapped random walk (few jumps per atom, as in a trajectory),
device resident float32 in and out.  usage: python scratch/time_unwrap.py [atoms] [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lammps_analysis_amd.engine as engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(1)
box = np.array([65.0, 65.0, 65.0])
free = np.cumsum(rng.normal(scale=0.05, size=(N, F, 3)).astype(np.float32), axis=1,
                 dtype=np.float32)
free += (rng.random((N, 1, 3)) * box).astype(np.float32)
wrapped = (free - np.floor(free / box) * box).astype(np.float32)
del free
x = engine.DeviceArray.from_numpy(wrapped)
y = engine.DeviceArray(wrapped.shape, np.float32)


def t(fn, n=5):
    """kernel time per call from the library's event timers"""
    fn(); engine.synchronize()
    engine.profile_enable(True); engine.profile_reset()
    for _ in range(n):
        fn()
    engine.synchronize()
    ms, launches = engine.profile_query("transform")
    engine.profile_enable(False)
    return ms * 1e-3 / max(launches, 1)


nbytes = 2 * wrapped.nbytes
for label, env in (("stream kernel", None), ("block-per-atom kernel", "1000000000")):
    if env:
        os.environ["MDX_UNWRAP_STREAM_MIN_ATOMS"] = env
    else:
        os.environ.pop("MDX_UNWRAP_STREAM_MIN_ATOMS", None)
    s = t(lambda: engine.unwrap_coordinates(x, box, out=y))
    print(f"{label}: {s*1e3:.3f} ms, {nbytes/s/1e12:.2f} TB/s (read + write float32)", flush=True)
    if env is None:
        first = y.to_numpy()
    else:
        assert np.array_equal(first, y.to_numpy())
for odd in (F - 1, F - 2):
    xs = engine.DeviceArray.from_numpy(np.ascontiguousarray(wrapped[:, :odd]))
    ys = engine.DeviceArray((N, odd, 3), np.float32)
    os.environ.pop("MDX_UNWRAP_STREAM_MIN_ATOMS", None)
    s = t(lambda: engine.unwrap_coordinates(xs, box, out=ys))
    print(f"stream kernel, {odd} frames (word-wise rows): {s*1e3:.3f} ms, "
          f"{2*xs.nbytes/s/1e12:.2f} TB/s", flush=True)
    xs.free(); ys.free()
