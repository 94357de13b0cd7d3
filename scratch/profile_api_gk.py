"""Where the wall time of project.run.GreenKuboIonicConductivity(data_range=500) goes for a
10k-atom trajectory in the on-disk store (IonicCurrent transformation included).
usage: python scratch/profile_api_gk.py [frames]"""
import cProfile, os, pstats, sys, tempfile, time, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lammps_analysis_amd as mds
from lammps_analysis_amd.database.mdsuite_properties import mdsuite_properties
from lammps_analysis_amd.database.simulation_database import (SpeciesInfo, TrajectoryChunkData,
                                                              TrajectoryMetadata)
from lammps_analysis_amd.file_io.file_read import FileProcessor

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
props = [mdsuite_properties.velocities]
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
                chunk.add_data(rng.standard_normal((nf, 5000, 3), dtype=np.float32), 0, sp.name,
                               "Velocities")
            yield chunk


work = tempfile.mkdtemp(prefix="mdx_prof_", dir=os.environ.get("TMPDIR") or None)
project = mds.Project(name="p", storage_path=work)
project.add_experiment("melt", timestep=0.002, temperature=1400.0, units="metal",
                       simulation_data=Synthetic())
# warm the library (code object, streams) with a small unrelated call
from lammps_analysis_amd import engine
engine.weighted_atom_sum(np.zeros((8, 8, 3), np.float32))
for label, kw in (("first call (with the IonicCurrent transformation)", {}),
                  ("second call (series stored)", {"data_range": 400})):
    prof = cProfile.Profile()
    t0 = time.perf_counter()
    prof.enable()
    project.run.GreenKuboIonicConductivity(plot=False, **({"data_range": 500} | kw))
    prof.disable()
    wall = time.perf_counter() - t0
    print(f"== {label}: {wall*1e3:.0f} ms wall, {n_frames/wall:.0f} frames/s")
    pstats.Stats(prof).sort_stats("cumulative").print_stats(40)
    pstats.Stats(prof).sort_stats("tottime").print_stats(14)
shutil.rmtree(work, ignore_errors=True)
