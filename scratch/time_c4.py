import time, numpy as np
import lammps_analysis_amd.engine as engine
sizes = [33334, 66666]; F = 32
dev = []
for s, n in enumerate(sizes):
    d = engine.DeviceArray((n, F, 3), np.float32); d.fill_uniform(seed=90 + s, lo=0.0, hi=100.0); dev.append(d)
eng = engine.RdfEngine(sizes, [100.0]*3, np.float32(49.9), 4990)
eng.accumulate(dev); engine.synchronize()
t0 = time.perf_counter(); eng.accumulate(dev); engine.synchronize(); dt = time.perf_counter() - t0
pairs = 99998 * 99997 / 2 * F
print(f"C4: {F/dt:.1f} frames/s, {pairs/dt:.3e} pairs/s")
for shards in (8,):
    t0 = time.perf_counter(); eng.accumulate(dev, shard_rank=3, shard_count=shards); engine.synchronize(); dt = time.perf_counter() - t0
    print(f"C4 one of {shards} tile shards: {dt*1e3:.1f} ms for {F} frames -> {F/dt:.1f} frames/s per node if all ranks alike")
