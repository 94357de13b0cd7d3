"""Multi-rank path on CPU: two, three or eight processes joined through the library's own host
transport (``lammps_analysis_amd.parallel``; no GPU here, so no RCCL).

Each rank takes its share of the units (windows / tiles / frames) exactly as the
calculators do and the exchange step must reproduce the single-process result.  The
partial results are NumPy here; ``tests/test_gpu_ranks.py`` runs the same paths through
the kernels.
"""
import numpy as np
import pytest

import _rank_workers as rw
from lammps_analysis_amd import parallel


@pytest.mark.parametrize("world", (2, 3, 8))
def test_exchange_step_matches_single_process(world):
    results = rw.run_ranks(rw.exchange, world)
    rng = np.random.default_rng(42)
    series = rng.normal(size=400)
    acfs = rw.window_acfs(series, 16, 301)
    integrals = np.cumsum(acfs, axis=1)[:, 1:]
    hist = np.zeros((3, 50), dtype=np.uint64)
    for t in range(37):
        hist[t % 3, (7 * t) % 50] += np.uint64(2**40 + t)
    frames_seen = []
    for rank, (total, n, mean, std, hist_total, frames, slowest, names, told, done) in results:
        np.testing.assert_allclose(total, acfs.sum(axis=0), rtol=1e-13)
        assert n == 301
        np.testing.assert_allclose(mean, integrals.mean(axis=0), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(std, integrals.std(axis=0), rtol=1e-10)
        np.testing.assert_array_equal(hist_total, hist)
        assert hist_total.dtype == np.uint64
        assert slowest == float(world)
        assert names == [f"rank{r}" for r in range(world)]
        assert told == {"from": 0} and done == "written"
        frames_seen += frames.tolist()
    assert frames_seen == list(range(11))
    # sums are formed in rank order on one rank: bit-identical everywhere
    for _, item in results[1:]:
        np.testing.assert_array_equal(item[0], results[0][1][0])


def test_writer_failure_reaches_every_rank():
    results = rw.run_ranks(rw.writer_failure, 2)
    assert results[0][1] == "writer:disk full"
    assert results[1][1].startswith("other:rank 0 failed: OSError: disk full")


def test_shared_project_has_one_writer_and_sharded_transformation(tmp_path):
    """Ingest and a multi-species transformation with two ranks in one project directory
    (mdsuite/transformations/transformations.py:553-619 is the loop being sharded): the
    stored series equals the single-process sum, each rank reduced only its frames, and no
    temporary file of a second writer is left behind."""
    results = rw.run_ranks(rw.sharded_transformation, 2, str(tmp_path))
    vel = rw.make_velocities()
    # stored arrays are float32; the transformation sees them widened
    expected = (vel["Na"].astype(np.float32).astype(np.float64).sum(axis=1)
                - vel["Cl"].astype(np.float32).astype(np.float64).sum(axis=1))
    for rank, (series, calls, files) in results:
        assert series.shape == (1, 257, 3)
        np.testing.assert_allclose(series[0], expected, rtol=0, atol=1e-5)
        assert not [f for f in files if f.endswith(".tmp")]
    # 257 frames: rank 0 takes 129, rank 1 takes 128, for each of the two species
    assert results[0][1][1] == [129, 129]
    assert results[1][1][1] == [128, 128]
    np.testing.assert_array_equal(results[0][1][0], results[1][1][0])


def test_split_range_covers_everything():
    for n in (0, 1, 7, 8, 99500):
        for size in (1, 2, 3, 8):
            parts = [parallel.split_range(n, r, size) for r in range(size)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_needs_no_transport():
    assert parallel.world() == (0, 1)
    assert parallel.transport() == "none" and parallel.communicator() is None
    x = np.arange(5.0)
    assert parallel.all_reduce_sum(x) is x
    assert parallel.run_on_writer(lambda: 7) == 7


@pytest.mark.parametrize("transport", ("auto", "rccl"))
def test_failed_rccl_setup_is_an_error_on_every_rank(transport):
    """Every rank owns a GPU (stubbed) and the communicator cannot be created on one of
    them: no rank may carry on with host sums, and none is left half-initialised."""
    results = rw.run_ranks(rw.rccl_setup_fails, 2, transport, raw=True)
    for rank, outcome in results:
        assert outcome.startswith("error:the RCCL communicator could not be created"), outcome
        assert "stub: ncclCommInitRank failed" in outcome
        assert outcome.endswith("|initialised=False|channel=None")


def test_host_transport_by_name_skips_rccl():
    results = rw.run_ranks(rw.rccl_setup_fails, 2, "host", raw=True)
    assert [o for _, o in results] == ["joined:host", "joined:host"]


def test_launcher_reports_the_failed_setup_as_exit_status(tmp_path):
    """The same failure through ``parallel.launch`` (what ``bench.py --gpus N`` uses): the
    launcher's status is non-zero."""
    script = tmp_path / "job.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {str(rw.ROOT)!r}); sys.path.insert(0, {str(rw.ROOT / 'tests')!r})\n"
        "import _rank_workers as rw\n"
        "from lammps_analysis_amd import parallel\n"
        "rank, world = int(os.environ['MDX_RANK']), int(os.environ['MDX_WORLD_SIZE'])\n"
        "rw.stub_gpu_per_rank(rank, world, create_fails_on=(0,))\n"
        "parallel.init()\n"
        "print('joined', parallel.transport())\n")
    assert parallel.launch([str(script)], 2, timeout=120) != 0
    env = dict(__import__("os").environ, MDX_COMM_TRANSPORT="host")
    assert parallel.launch([str(script)], 2, env=env, timeout=120) == 0


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_tcp_rendezvous_needs_the_shared_token():
    address = f"tcp:127.0.0.1:{_free_port()}"
    results = rw.run_ranks(rw.token_exchange, 2, "s3cret", address=address, raw=True)
    assert [o for _, o in results] == ["sum=3", "sum=3"]
    # no token at all: refused before anything is read from the wire
    address = f"tcp:127.0.0.1:{_free_port()}"
    results = rw.run_ranks(rw.token_exchange, 2, None, address=address, raw=True)
    for _, outcome in results:
        assert outcome.endswith("|initialised=False")
        assert outcome.split("|")[0] in ("PermissionError", "TimeoutError", "ConnectionError")


def test_tcp_rendezvous_drops_a_peer_with_the_wrong_token():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    address = f"tcp:127.0.0.1:{_free_port()}"
    procs = [ctx.Process(target=rw.token_exchange, args=(0, 2, address, queue, "right")),
             ctx.Process(target=rw.token_exchange, args=(1, 2, address, queue, "wrong"))]
    for p in procs:
        p.start()
    outcomes = dict(queue.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    # rank 1 is refused (or sees the server hang up); rank 0 never unpickles anything from
    # it and gives up when the rendezvous time is over
    assert outcomes[1].split("|")[0] in ("PermissionError", "ConnectionError")
    assert outcomes[0].split("|")[0] == "TimeoutError"
    assert all(o.endswith("|initialised=False") for o in outcomes.values())
