"""Host-input RDF over a long trajectory: frames/s from anonymous memory and from a
memory-mapped .npy file, for several staging block sizes and gather thread counts
(MDX_RDF_HOST_BLOCK / MDX_RDF_GATHER_THREADS are read at every call).
usage: python scratch/time_host_rdf_long.py [frames]"""
import os, sys, tempfile, time, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lammps_analysis_amd.engine as engine

N = 5000
F = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
work = tempfile.mkdtemp(prefix="mdx_host_", dir=os.environ.get("TMPDIR") or None)
rng = np.random.default_rng(0)
maps = []
for s in range(2):
    m = np.lib.format.open_memmap(os.path.join(work, f"s{s}.npy"), mode="w+",
                                  dtype=np.float32, shape=(N, F, 3))
    for a0 in range(0, N, 500):
        m[a0:a0 + 500] = rng.random((500, F, 3), dtype=np.float32) * np.float32(65.0)
    m.flush()
    del m
eng = engine.RdfEngine([N, N], [65.0] * 3, np.float32(32.4), 3240)


def run(arrays, label):
    eng.reset()
    engine.synchronize()
    t0 = time.perf_counter()
    eng.accumulate(arrays)
    engine.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:60s} {F / dt:9.0f} frames/s  {dt * 1e3:8.1f} ms", flush=True)
    return eng.read().copy()


def fresh_maps():
    return [np.load(os.path.join(work, f"s{s}.npy"), mmap_mode="r") for s in range(2)]


dev = [engine.DeviceArray.from_numpy(np.ascontiguousarray(m[:, :min(F, 8192)])) for m in fresh_maps()]
eng.accumulate(dev); engine.synchronize()
t0 = time.perf_counter(); eng.accumulate(dev); engine.synchronize()
print(f"device-resident {min(F, 8192) / (time.perf_counter() - t0):.0f} frames/s", flush=True)
del dev

ref = None
for block, threads in ((0, 0), (512, 2), (512, 4), (512, 8), (512, 16), (256, 8), (1024, 8),
                       (2048, 8), (1024, 16)):
    for k, v in (("MDX_RDF_HOST_BLOCK", block), ("MDX_RDF_GATHER_THREADS", threads)):
        if v:
            os.environ[k] = str(v)
        else:
            os.environ.pop(k, None)
    h = run(fresh_maps(), f"memmap (fresh mapping) block={block or 'default'} threads={threads or 'default'}")
    if ref is None:
        ref = h
    assert np.array_equal(ref, h)
os.environ.pop("MDX_RDF_HOST_BLOCK", None)
os.environ.pop("MDX_RDF_GATHER_THREADS", None)
ram = [np.array(m) for m in fresh_maps()]
run(ram, "anonymous memory, defaults")
os.environ["MDX_RDF_GATHER_THREADS"] = "8"
run(ram, "anonymous memory, threads=8")
shutil.rmtree(work, ignore_errors=True)
