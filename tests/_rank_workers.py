"""Bodies of the multi-rank tests: each function runs in a process of its own (spawn),
joins the job through ``lammps_analysis_amd.parallel`` and puts its results on a queue."""
import os
import pathlib
import sys
import traceback

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def run_ranks(target, world, *args, timeout=240, transport=None, address=None, raw=False):
    """Start ``world`` processes that join the job and run ``target(rank, world, *args)``;
    returns their queue items ordered by rank.  A rank that raises reports the traceback.
    ``raw``: the process runs ``target(rank, world, address, queue, *args)`` and joins (or
    fails to join) the job itself."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    if address is None:
        address = f"unix:mdx-test-{os.getpid()}-{np.random.default_rng().integers(1 << 62)}"
    if raw:
        procs = [ctx.Process(target=target, args=(r, world, address, queue, *args))
                 for r in range(world)]
    else:
        procs = [ctx.Process(target=_guarded,
                             args=(target, r, world, address, queue, args, transport))
                 for r in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in procs:
            results.append(queue.get(timeout=timeout))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for item in results:
        if isinstance(item[1], str) and item[1].startswith("Traceback"):
            raise AssertionError(f"rank {item[0]} failed:\n{item[1]}")
    return sorted(results, key=lambda r: r[0])


def _guarded(target, rank, world, address, queue, args, transport=None):
    try:
        from lammps_analysis_amd import parallel
        parallel.init(rank=rank, world_size=world, local_rank=rank, address=address,
                      transport=transport)
        queue.put((rank, target(rank, world, *args)))
        parallel.barrier()
        parallel.shutdown()
    except BaseException:
        queue.put((rank, traceback.format_exc()))


def window_acfs(series, d, n_win):
    out = []
    for w in range(n_win):
        win = series[w:w + d]
        out.append(np.array([np.sum(win[:d - m] * win[m:]) / (d - m) for m in range(d)]))
    return np.array(out)


def exchange(rank, world):
    """The exchange step on NumPy partials: sums, maxima, objects, moments, histograms."""
    from lammps_analysis_amd import parallel
    from lammps_analysis_amd.engine import Moments

    assert parallel.world() == (rank, world)
    assert parallel.transport() == "host"  # no GPU per rank in this test
    assert parallel.is_writer() == (rank == 0)
    rng = np.random.default_rng(42)  # same data on every rank
    series = rng.normal(size=400)
    d, n_win = 16, 301
    begin, end = parallel.split_range(n_win, rank, world)
    acfs = window_acfs(series, d, n_win)[begin:end]
    integrals = np.cumsum(acfs, axis=1)[:, 1:]
    mean = integrals.mean(axis=0)
    moments = Moments(end - begin, mean, ((integrals - mean) ** 2).sum(axis=0))
    total = parallel.all_reduce_sum(acfs.sum(axis=0))
    merged = parallel.merge_moments(moments)
    n_tiles = 37  # integer histogram, tiles dealt cyclically like mdx_rdf_accumulate does
    hist = np.zeros((3, 50), dtype=np.uint64)
    for t in range(n_tiles):
        if t % world == rank:
            hist[t % 3, (7 * t) % 50] += np.uint64(2**40 + t)
    hist_total = parallel.all_reduce_sum(hist)
    frames = parallel.split_frames(np.arange(11), rank, world)
    slowest = float(parallel.all_reduce_max([1.0 + rank])[0])
    names = parallel.all_gather_objects(f"rank{rank}")
    told = parallel.broadcast_object({"from": rank})
    done = parallel.run_on_writer(lambda: "written")
    return (total, merged.n, merged.mean, merged.std(), hist_total, frames, slowest, names,
            told, done)


def writer_failure(rank, world):
    """A failure on the writing rank reaches every rank instead of hanging them."""
    from lammps_analysis_amd import parallel

    def boom():
        raise OSError("disk full")

    try:
        parallel.run_on_writer(boom)
    except OSError as exc:
        return f"writer:{exc}"
    except RuntimeError as exc:
        return f"other:{exc}"
    return "no error"


class _NumpyCurrent:
    """IonicCurrent with a NumPy body, so the sharded transformation loop runs on CPU."""

    @staticmethod
    def build():
        from lammps_analysis_amd.database.mdsuite_properties import mdsuite_properties
        from lammps_analysis_amd.transformations.transformations import MultiSpeciesTrafo

        class NumpyIonicCurrent(MultiSpeciesTrafo):
            calls = []

            def __init__(self):
                super().__init__(
                    input_properties=[mdsuite_properties.velocities, mdsuite_properties.charge],
                    output_property=mdsuite_properties.ionic_current,
                    scale_function={"linear": {"scale_factor": 2}})

            def transform_batch(self, batch, carryover=None):
                out = 0.0
                for props in batch.values():
                    vel = np.asarray(props["Velocities"], dtype=np.float64)
                    NumpyIonicCurrent.calls.append(vel.shape[1])
                    out = out + (props["Charge"] * vel).sum(axis=0)
                return out

        return NumpyIonicCurrent


def make_velocities(n_step=257, n_na=7, n_cl=5, seed=3):
    rng = np.random.default_rng(seed)
    return {"Na": rng.normal(size=(n_step, n_na, 3)), "Cl": rng.normal(size=(n_step, n_cl, 3))}


def add_velocities(mds, exp, vel, box_l=None):
    from lammps_analysis_amd.database.mdsuite_properties import mdsuite_properties
    from lammps_analysis_amd.database.simulation_database import (SpeciesInfo,
                                                                  TrajectoryChunkData,
                                                                  TrajectoryMetadata)
    from lammps_analysis_amd.file_io.script_input import ScriptInput

    charges = {"Na": 1.0, "Cl": -1.0}
    n_step = next(iter(vel.values())).shape[0]
    species = [SpeciesInfo(name=k, n_particles=v.shape[1],
                           properties=[mdsuite_properties.velocities], charge=charges[k])
               for k, v in vel.items()]
    metadata = TrajectoryMetadata(species_list=species, n_configurations=n_step,
                                  sample_rate=1, box_l=box_l)
    data = TrajectoryChunkData(species_list=species, chunk_size=n_step)
    for k, v in vel.items():
        data.add_data(v, 0, k, "Velocities")
    exp.add_data(ScriptInput(data=data, metadata=metadata, name="test_name"))


def sharded_transformation(rank, world, workdir):
    """Project + ingest + a frame-sharded multi-species transformation in one shared
    directory: one writer, every rank ends with the same stored series."""
    import lammps_analysis_amd as mds

    os.chdir(workdir)
    project = mds.Project(name="shared")
    exp = project.add_experiment("nacl", timestep=0.002, temperature=1400.0, units="metal")
    vel = make_velocities()
    add_velocities(mds, exp, vel, box_l=[20.0, 20.0, 20.0])
    trafo_cls = _NumpyCurrent.build()
    exp.cls_transformation_run(trafo_cls())
    from lammps_analysis_amd import parallel
    parallel.barrier()
    from lammps_analysis_amd.database.simulation_database import Database
    db = Database(exp.database_path / "database.hdf5")
    series = np.array(db.load_data(["Observables/Ionic_Current"])["Observables/Ionic_Current"])
    return series, list(trafo_cls.calls), sorted(os.listdir(project.project_dir))


def gpu_calculators(rank, world, workdir):
    """RDF and Green-Kubo ionic conductivity through the product API with the ranks
    sharing GPU 0: tile shards / frame shards / window shards through the real kernels."""
    import lammps_analysis_amd as mds
    from lammps_analysis_amd import parallel
    from lammps_analysis_amd.database.mdsuite_properties import mdsuite_properties
    from lammps_analysis_amd.database.simulation_database import (SpeciesInfo,
                                                                  TrajectoryChunkData,
                                                                  TrajectoryMetadata)
    from lammps_analysis_amd.file_io.script_input import ScriptInput

    os.chdir(workdir)
    rng = np.random.default_rng(11)
    n_step, box = 700, 11.0
    pos = {"Na": rng.uniform(0, box, size=(n_step, 300, 3)),
           "Cl": rng.uniform(0, box, size=(n_step, 280, 3))}
    vel = {k: rng.normal(size=v.shape) for k, v in pos.items()}
    project = mds.Project(name=f"ranks{world}")
    exp = project.add_experiment("nacl", timestep=0.002, temperature=1400.0, units="metal")
    charges = {"Na": 1.0, "Cl": -1.0}
    species = [SpeciesInfo(name=k, n_particles=v.shape[1],
                           properties=[mdsuite_properties.positions,
                                       mdsuite_properties.velocities], charge=charges[k])
               for k, v in pos.items()]
    metadata = TrajectoryMetadata(species_list=species, n_configurations=n_step,
                                  sample_rate=1, box_l=[box] * 3)
    data = TrajectoryChunkData(species_list=species, chunk_size=n_step)
    for k in pos:
        data.add_data(pos[k], 0, k, "Positions")
        data.add_data(vel[k], 0, k, "Velocities")
    exp.add_data(ScriptInput(data=data, metadata=metadata, name="test_name"))
    rdf = exp.run.RadialDistributionFunction(number_of_configurations=200, plot=False)
    rdf_tiles = exp.run.RadialDistributionFunction(number_of_configurations=150, plot=False,
                                                   shard="tiles")
    rdf_frames = exp.run.RadialDistributionFunction(number_of_configurations=151, plot=False,
                                                    shard="frames")
    gk = exp.run.GreenKuboIonicConductivity(plot=False, data_range=120, correlation_time=1)
    vacf = exp.run.GreenKuboDiffusionCoefficients(plot=False, data_range=60,
                                                  correlation_time=1)
    parallel.barrier()  # the writing rank has finished the result file
    out = {
        "transport": parallel.transport(),
        "rdf": {k: np.array(v["y"][1:]) for k, v in rdf.data_dict.items()},
        "rdf_tiles": {k: np.array(v["y"][1:]) for k, v in rdf_tiles.data_dict.items()},
        "rdf_frames": {k: np.array(v["y"][1:]) for k, v in rdf_frames.data_dict.items()},
        "gk_acf": np.array(gk["System"]["acf"]),
        "gk_sigma": gk["System"]["ionic_conductivity"],
        "gk_unc": gk["System"]["uncertainty"],
        "vacf": {k: np.array(v["acf"]) for k, v in vacf.data_dict.items()},
        "files": sorted(os.listdir(project.project_dir)),
    }
    return out


def rccl_single_rank(rank, world):
    """A communicator of one rank: the device collective without a second GPU."""
    from lammps_analysis_amd import engine, parallel

    rng = np.random.default_rng(5)
    counts = rng.integers(0, 2**62, size=1000, dtype=np.uint64)
    values = rng.normal(size=777)
    out = {"transport": parallel.transport()}
    # world() == 1 short-cuts the Python collectives, so call the library directly
    from lammps_analysis_amd import _lib
    lib, comm = _lib.load(), parallel.communicator()
    work = counts.copy()
    _lib.check(lib.mdx_comm_allreduce(comm, work.ctypes.data, _lib.MDX_HOST, work.size,
                                      _lib.MDX_COMM_U64, _lib.MDX_COMM_SUM))
    out["u64_ok"] = bool(np.array_equal(work, counts))
    fwork = values.copy()
    _lib.check(lib.mdx_comm_allreduce(comm, fwork.ctypes.data, _lib.MDX_HOST, fwork.size,
                                      _lib.MDX_COMM_F64, _lib.MDX_COMM_SUM))
    out["f64_ok"] = bool(np.array_equal(fwork, values))
    dev = engine.DeviceArray.from_numpy(values)
    _lib.check(lib.mdx_comm_allreduce(comm, dev.ptr, _lib.MDX_DEVICE, values.size,
                                      _lib.MDX_COMM_F64, _lib.MDX_COMM_MAX))
    out["max_ok"] = bool(np.array_equal(dev.to_numpy(), values))
    pos = [rng.uniform(0, 9, size=(40, 3, 3)).astype(np.float32) for _ in range(2)]
    rdf = engine.RdfEngine([40, 40], [9.0] * 3, 4.4, 440)
    rdf.accumulate(pos)
    before = rdf.read()
    rdf.reduce(comm)
    out["rdf_ok"] = bool(np.array_equal(rdf.read(), before) and before.sum() > 0)
    rdf.close()
    return out


def stub_gpu_per_rank(rank, world, create_fails_on=()):
    """Makes this process look like one of ``world`` ranks that each own a GPU with a
    loadable librccl, without touching a device: the rendezvous then takes the RCCL branch.
    ``mdx_comm_create`` fails on the listed ranks."""
    from lammps_analysis_amd import _lib

    real = _lib.load()

    class Stub:
        def __getattr__(self, name):
            return getattr(real, name)

        def mdx_set_device(self, device):
            return 0

        def mdx_comm_available(self):
            return 1

        def mdx_comm_unique_id(self, buf, n):
            return 0

        def mdx_comm_create(self, handle, r, n, uid, n_bytes):
            return -2 if r in create_fails_on else 0

        def mdx_comm_destroy(self, handle):
            return 0

        def mdx_last_error(self):
            return b"stub: ncclCommInitRank failed"

    stub = Stub()
    _lib.load = lambda: stub
    _lib.device_count = lambda: world


def rccl_setup_fails(rank, world, address, queue, transport):
    """Joins a job whose communicator cannot be created on rank 1 and reports what
    ``parallel.init`` did about it."""
    try:
        from lammps_analysis_amd import parallel
        stub_gpu_per_rank(rank, world, create_fails_on=(1,))
        try:
            parallel.init(rank=rank, world_size=world, local_rank=rank, address=address,
                          transport=transport)
            outcome = "joined:" + parallel.transport()
            parallel.barrier()
            parallel.shutdown()
        except RuntimeError as exc:
            outcome = f"error:{exc}|initialised={parallel._state.initialised}" \
                      f"|channel={parallel._state.channel}"
        queue.put((rank, outcome))
    except BaseException:
        queue.put((rank, traceback.format_exc()))


def token_exchange(rank, world, address, queue, token):
    """A tcp: rendezvous with this rank's idea of the shared secret."""
    try:
        from lammps_analysis_amd import parallel
        if token is None:
            os.environ.pop("MDX_COMM_TOKEN", None)
        else:
            os.environ["MDX_COMM_TOKEN"] = token
        os.environ["MDX_COMM_TIMEOUT"] = "6"
        import importlib
        importlib.reload(parallel)
        try:
            parallel.init(rank=rank, world_size=world, local_rank=rank, address=address)
            total = parallel.all_reduce_sum(np.array([rank + 1.0]))
            outcome = f"sum={total[0]:.0f}"
            parallel.barrier()
            parallel.shutdown()
        except (PermissionError, TimeoutError, ConnectionError) as exc:
            outcome = f"{type(exc).__name__}|initialised={parallel._state.initialised}"
        queue.put((rank, outcome))
    except BaseException:
        queue.put((rank, traceback.format_exc()))
