"""Multi-rank path on CPU: two processes, gloo backend.

Each rank takes its share of the units (windows / tiles / frames) exactly as the
calculators do, produces partial results, and the exchange step (one all-reduce of
sums, one merge of window statistics) must reproduce the single-process result.
The partial results are computed with NumPy here, the GPU kernels are covered by
the gpu-marked tests.
"""
import os
import socket
import sys
import pathlib

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _window_integrals(series, d, n_win):
    out = []
    for w in range(n_win):
        win = series[w:w + d]
        acf = np.array([np.sum(win[:d - m] * win[m:]) / (d - m) for m in range(d)])
        out.append(acf)
    return np.array(out)


def _worker(rank, world, port, queue):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist
    from lammps_analysis_amd import parallel
    from lammps_analysis_amd.engine import Moments

    assert parallel.world() == (0, 1)            # before init: single process
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert parallel.world() == (rank, world)
        rng = np.random.default_rng(42)             # same data on every rank
        series = rng.normal(size=400)
        d, n_win = 16, 301
        begin, end = parallel.split_range(n_win, rank, world)
        acfs = _window_integrals(series, d, n_win)[begin:end]
        integrals = np.cumsum(acfs, axis=1)[:, 1:]
        local_sum = acfs.sum(axis=0)
        mean = integrals.mean(axis=0)
        moments = Moments(end - begin, mean, ((integrals - mean) ** 2).sum(axis=0))
        total = parallel.all_reduce_sum(local_sum)
        merged = parallel.merge_moments(moments)
        # integer histogram, tiles dealt cyclically like mdx_rdf_accumulate does
        n_tiles = 37
        hist = np.zeros((3, 50), dtype=np.uint64)
        for t in range(n_tiles):
            if t % world == rank:
                hist[t % 3, (7 * t) % 50] += np.uint64(2**40 + t)
        hist_total = parallel.all_reduce_sum(hist)
        frames = parallel.split_frames(np.arange(11), rank, world)
        queue.put((rank, total, merged.n, merged.mean, merged.std(), hist_total, frames))
    finally:
        dist.destroy_process_group()


def test_two_rank_exchange_matches_single_process():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    rng = np.random.default_rng(42)
    series = rng.normal(size=400)
    acfs = _window_integrals(series, 16, 301)
    integrals = np.cumsum(acfs, axis=1)[:, 1:]
    hist = np.zeros((3, 50), dtype=np.uint64)
    for t in range(37):
        hist[t % 3, (7 * t) % 50] += np.uint64(2**40 + t)
    frames_seen = []
    for rank, total, n, mean, std, hist_total, frames in sorted(results, key=lambda r: r[0]):
        np.testing.assert_allclose(total, acfs.sum(axis=0), rtol=1e-13)
        assert n == 301
        np.testing.assert_allclose(mean, integrals.mean(axis=0), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(std, integrals.std(axis=0), rtol=1e-10)
        np.testing.assert_array_equal(hist_total, hist)
        frames_seen += frames.tolist()
    assert frames_seen == list(range(11))


def test_split_range_covers_everything():
    from lammps_analysis_amd import parallel
    for n in (0, 1, 7, 8, 99500):
        for size in (1, 2, 3, 8):
            parts = [parallel.split_range(n, r, size) for r in range(size)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1
