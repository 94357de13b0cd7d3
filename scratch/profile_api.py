"""Where the wall time of project.run.RadialDistributionFunction(number_of_configurations=-1)
goes for a trajectory in the on-disk store: cProfile of the call plus timers around the
engine calls.  usage: python scratch/profile_api.py [frames]"""
import cProfile, os, pstats, sys, tempfile, time, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lammps_analysis_amd as mds
from lammps_analysis_amd import engine
from lammps_analysis_amd.database.mdsuite_properties import mdsuite_properties
from lammps_analysis_amd.database.simulation_database import (SpeciesInfo, TrajectoryChunkData,
                                                              TrajectoryMetadata)
from lammps_analysis_amd.file_io.file_read import FileProcessor

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
props = [mdsuite_properties.positions]
species = [SpeciesInfo(name=n, n_particles=5000, properties=props, charge=q)
           for n, q in (("Na", 1.0), ("Cl", -1.0))]
meta = TrajectoryMetadata(species_list=species, n_configurations=n_frames, sample_rate=1,
                          box_l=[65.0] * 3)
rng = np.random.default_rng(7)


class Synthetic(FileProcessor):
    def __str__(self):
        return "synthetic"

    def _get_metadata(self):
        return meta

    def get_configurations_generator(self):
        for f0 in range(0, n_frames, 500):
            nf = min(500, n_frames - f0)
            chunk = TrajectoryChunkData(species_list=species, chunk_size=nf)
            for sp in species:
                chunk.add_data(rng.random((nf, 5000, 3), dtype=np.float32) * np.float32(65.0),
                               0, sp.name, "Positions")
            yield chunk


work = tempfile.mkdtemp(prefix="mdx_prof_", dir=os.environ.get("TMPDIR") or None)
project = mds.Project(name="p", storage_path=work)
project.add_experiment("melt", timestep=0.002, temperature=1400.0, units="metal",
                       simulation_data=Synthetic())
timers = {}
for name in ("accumulate", "read", "__init__", "close"):
    orig = getattr(engine.RdfEngine, name)

    def wrapped(self, *a, _orig=orig, _name=name, **k):
        t0 = time.perf_counter()
        out = _orig(self, *a, **k)
        if _name == "accumulate":
            engine.synchronize()
        timers[_name] = timers.get(_name, 0.0) + time.perf_counter() - t0
        return out
    setattr(engine.RdfEngine, name, wrapped)
for label, kw in (("first call", {}), ("second call (other bins)", {"number_of_bins": 3000})):
    timers.clear()
    prof = cProfile.Profile()
    t0 = time.perf_counter()
    prof.enable()
    project.run.RadialDistributionFunction(number_of_configurations=-1, plot=False, **kw)
    prof.disable()
    wall = time.perf_counter() - t0
    print(f"== {label}: {wall*1e3:.0f} ms wall, {(n_frames-1)/wall:.0f} frames/s; engine timers (ms):",
          {k: round(v * 1e3, 1) for k, v in timers.items()})
    pstats.Stats(prof).sort_stats("cumulative").print_stats(18)
shutil.rmtree(work, ignore_errors=True)
