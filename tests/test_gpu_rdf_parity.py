"""GPU parity of the RDF histogram kernels against the CPU oracle (bit-exact)."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import rdf as orc

pytestmark = pytest.mark.gpu


def _positions(rng, n_per_species, n_frames, box):
    return [rng.uniform(0.0, 1.0, (n, n_frames, 3)).astype(np.float32)
            * np.asarray(box, dtype=np.float32) for n in n_per_species]


def _gpu_hist(engine, pos, n_per_species, box, cutoff, nbins, strict=True, literal=False,
              frames=None, device=False, shards=1):
    eng = engine.RdfEngine(n_per_species, box, cutoff, nbins,
                           strict_reference_masks=strict, literal_kernel=literal)
    data = [engine.DeviceArray.from_numpy(p) for p in pos] if device else pos
    for r in range(shards):
        eng.accumulate(data, frames=frames, shard_rank=r, shard_count=shards)
    hist = eng.read()
    eng.close()
    return hist


@pytest.mark.parametrize("literal", [True, False])
@pytest.mark.parametrize("strict", [True, False])
def test_small_two_species_matches_numpy_oracle(gpu, literal, strict):
    rng = np.random.default_rng(0)
    n_per_species = [64, 64]
    box = [12.0, 12.0, 12.0]
    cutoff = np.float32(12.0 / 2 - 0.1)
    nbins = int(cutoff / 0.01)
    pos = _positions(rng, n_per_species, 4, box)
    got = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, strict, literal)
    ref_c = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins,
                              strict=strict)
    np.testing.assert_array_equal(got, ref_c)
    if strict:
        ref = orc.rdf_histograms(pos, ["Na", "Cl"], box, cutoff, nbins, rdf_minibatch=50)
        for i, key in enumerate(orc.key_names(["Na", "Cl"])):
            np.testing.assert_array_equal(got[i], ref[key].astype(np.uint64))


@pytest.mark.parametrize("literal", [True, False])
def test_ragged_three_species_rectangular_box(gpu, literal):
    rng = np.random.default_rng(1)
    n_per_species = [300, 1, 517]          # crosses tile boundaries; a 1-atom species
    box = [9.0, 11.5, 14.25]
    cutoff = np.float32(4.4)
    nbins = 173
    pos = _positions(rng, n_per_species, 3, box)
    got = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, True, literal)
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
    np.testing.assert_array_equal(got, ref)


def test_unwrapped_coordinates_and_large_cutoff(gpu):
    """Coordinates outside the box (several images away) and cutoff > L/2."""
    rng = np.random.default_rng(2)
    n_per_species = [200, 150]
    box = [8.0, 8.0, 8.0]
    pos = [(rng.uniform(-3.0, 4.0, (n, 2, 3)) * 8.0).astype(np.float32) for n in n_per_species]
    cutoff = np.float32(6.5)
    nbins = 650
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
    for literal in (True, False):
        got = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, True, literal)
        np.testing.assert_array_equal(got, ref)


def test_edge_placed_pairs(gpu):
    """Pairs placed on bin edges, at the cutoff and at half the box (rint ties)."""
    box = np.array([10.0, 10.0, 10.0], dtype=np.float32)
    cutoff = np.float32(4.9)
    nbins = 490
    step = float(cutoff) / nbins
    base = np.zeros((1, 3), dtype=np.float32)
    others = []
    for k in range(0, nbins + 3):
        for eps in (-1, 0, 1):
            d = np.nextafter(np.float32(k * step), np.float32(np.inf if eps > 0 else -np.inf)) \
                if eps else np.float32(k * step)
            others.append([d, 0.0, 0.0])
    for d in (5.0, np.nextafter(np.float32(5.0), np.float32(0)), np.nextafter(np.float32(5.0), np.float32(9)),
              -5.0, 15.0, 25.0, 4.9, np.nextafter(np.float32(4.9), np.float32(0))):
        others.append([0.0, d, 0.0])
    others = np.asarray(others, dtype=np.float32)
    # species A: one dummy first atom (dropped by the strict masks) + the origin atom
    a = np.concatenate([np.full((1, 3), 3.3, np.float32), base])[:, None, :]
    b = np.concatenate([np.full((1, 3), 7.7, np.float32), others])[:, None, :]
    pos = [np.ascontiguousarray(a), np.ascontiguousarray(b)]
    n_per_species = [a.shape[0], b.shape[0]]
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
    ref_np = orc.rdf_histograms(pos, ["A", "B"], box, cutoff, nbins)
    for i, key in enumerate(orc.key_names(["A", "B"])):
        np.testing.assert_array_equal(ref[i], ref_np[key].astype(np.uint64))
    for literal in (True, False):
        got = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, True, literal)
        np.testing.assert_array_equal(got, ref)


def test_frame_selection_device_input_and_sharding(gpu):
    rng = np.random.default_rng(3)
    n_per_species = [257, 255]
    box = [10.0, 10.0, 10.0]
    cutoff = np.float32(4.9)
    nbins = 490
    pos = _positions(rng, n_per_species, 9, box)
    frames = np.array([0, 2, 3, 8, 8], dtype=np.int64)
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins,
                            frames=frames)
    got_host = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, frames=frames)
    got_dev = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, frames=frames,
                        device=True, shards=3)
    np.testing.assert_array_equal(got_host, ref)
    np.testing.assert_array_equal(got_dev, ref)


def test_float64_input_is_rounded_like_tf_cast(gpu):
    rng = np.random.default_rng(4)
    n_per_species = [130]
    box = [7.0, 7.0, 7.0]
    pos64 = [rng.uniform(0.0, 7.0, (130, 2, 3))]
    cutoff = np.float32(3.4)
    nbins = 340
    ref = c_oracle.rdf_hist(pos64[0].astype(np.float32), n_per_species, box, cutoff, nbins)
    got = _gpu_hist(gpu, pos64, n_per_species, box, cutoff, nbins)
    np.testing.assert_array_equal(got, ref)


def test_empty_and_degenerate_inputs(gpu):
    eng = gpu.RdfEngine([1, 1], [5.0, 5.0, 5.0], 2.4, 24)     # strict masks leave no atoms
    eng.accumulate([np.zeros((1, 2, 3), np.float32), np.ones((1, 2, 3), np.float32)])
    assert eng.read().sum() == 0
    eng.close()
    with pytest.raises(ValueError):
        gpu.RdfEngine([4], [5.0, 5.0, 5.0], -1.0, 24)
    with pytest.raises(ValueError):
        gpu.RdfEngine([4], [5.0, 0.0, 5.0], 2.0, 24)


def test_medium_system_table_equals_literal(gpu):
    """2 x 1500 atoms, 6 frames: the table kernel against the literal kernel and the
    C oracle (about 2.7e7 pairs)."""
    rng = np.random.default_rng(5)
    n_per_species = [1500, 1500]
    box = [40.0, 40.0, 40.0]
    cutoff = np.float32(19.9)
    nbins = 1990
    pos = _positions(rng, n_per_species, 6, box)
    lit = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, literal=True)
    tab = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, literal=False)
    np.testing.assert_array_equal(tab, lit)
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
    np.testing.assert_array_equal(lit, ref)


def test_mixed_wrapped_and_unwrapped_frames(gpu):
    """Frames inside the box take the threshold minimum-image path, frames with
    atoms several images away the general one; both in one call."""
    rng = np.random.default_rng(6)
    n_per_species = [270, 300]
    box = [10.0, 12.0, 9.0]
    pos = _positions(rng, n_per_species, 6, box)
    for p in pos:
        p[:, 1] += np.float32(23.0) * rng.integers(-2, 3, size=(p.shape[0], 3)).astype(np.float32)
        p[5, 4, 0] = np.float32(1.0e4)
    cutoff = np.float32(4.4)
    nbins = 440
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
    got = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins)
    np.testing.assert_array_equal(got, ref)


def test_box_edge_coordinates_half_box_ties(gpu):
    """Atoms exactly half a box apart (rint ties) and exactly at 0 / L."""
    box = [8.0, 8.0, 8.0]
    grid = np.array([0.0, 4.0, 8.0, np.nextafter(np.float32(4.0), np.float32(0)),
                     np.nextafter(np.float32(4.0), np.float32(9)), 2.0, 6.0], dtype=np.float32)
    pts = np.array([[x, y, z] for x in grid for y in grid for z in grid], dtype=np.float32)
    pos = [np.ascontiguousarray(np.concatenate([np.zeros((1, 3), np.float32), pts])[:, None, :])]
    n_per_species = [pos[0].shape[0]]
    cutoff = np.float32(7.5)
    nbins = 300
    ref = c_oracle.rdf_hist(pos[0], n_per_species, box, cutoff, nbins)
    for literal in (True, False):
        got = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, literal=literal)
        np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_host_input_is_streamed_in_blocks(gpu, dtype):
    """Host arrays with 128 or more contiguous frames are copied block by block on a
    second stream while the previous block is histogrammed (two alternating staging
    sets, a ragged last block); the counts equal those of the one-shot paths."""
    rng = np.random.default_rng(31)
    n_per_species = [150, 0, 100]
    box = [10.0, 11.0, 12.0]
    cutoff, nbins = np.float32(4.9), 490
    n_frames = 333
    pos = [p.astype(dtype) for p in _positions(rng, n_per_species, n_frames, box)]
    streamed = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins)
    resident = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins, device=True)
    listed = _gpu_hist(gpu, pos, n_per_species, box, cutoff, nbins,
                       frames=np.arange(n_frames, dtype=np.int64))
    np.testing.assert_array_equal(streamed, resident)
    np.testing.assert_array_equal(streamed, listed)
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0).astype(np.float32), n_per_species, box,
                            cutoff, nbins)
    np.testing.assert_array_equal(streamed, ref)


@pytest.mark.parametrize("literal", [True, False])
@pytest.mark.parametrize("strict", [True, False])
def test_cubic_lattice_shell_counts(gpu, literal, strict):
    """Known answer independent of the oracle: a periodic simple cubic lattice has 6, 12,
    8, 6, 24, ... neighbours per shell, i.e. exact pair counts per distance."""
    from test_oracle_reference_pins import _cubic_lattice, lattice_histogram
    pos, box = _cubic_lattice()
    cutoff, nbins = np.float32(4.9), 490
    got = _gpu_hist(gpu, [pos], [len(pos)], box, cutoff, nbins, strict, literal)[0]
    full = lattice_histogram(nbins, cutoff).astype(np.int64)
    expected = full - full // 256 if strict else full
    np.testing.assert_array_equal(got.astype(np.int64), expected)
