"""Seeded random sweeps: many small configurations per kernel family against the oracle.

The parametrised parity tests pick their cases by hand; these draw sizes, boxes, cutoffs,
window layouts and dtypes at random (fixed seeds), so that combinations nobody thought of
get exercised too.  Bars as elsewhere: histograms bit-exact, float series within 1e-6 of the
largest magnitude.
"""
import numpy as np
import pytest
from scipy.integrate import cumulative_trapezoid

from oracle import c_oracle

pytestmark = pytest.mark.gpu


def _close(got, ref, rtol=1e-6):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    scale = max(float(np.max(np.abs(ref))) if ref.size else 1.0, 1e-300)
    np.testing.assert_allclose(got, ref, rtol=0.0, atol=rtol * scale)


@pytest.mark.parametrize("seed", range(40))
def test_rdf_random_configurations(gpu, seed):
    rng = np.random.default_rng(1000 + seed)
    n_species = int(rng.integers(1, 4))
    sizes = [int(rng.choice([0, 1, 2, 7, 63, 64, 65, 200, 257, 400])) for _ in range(n_species)]
    if sum(sizes) < 2:
        sizes[0] = 70
    box = [float(b) for b in rng.uniform(4.0, 20.0, 3)]
    if rng.random() < 0.4:
        box = [box[0]] * 3
    # below, at and well above half the shortest edge
    cutoff = np.float32(min(box) * rng.choice([0.2, 0.45, 0.5, 0.8]))
    nbins = int(rng.choice([1, 2, 17, 100, 999, 3000]))
    n_frames = int(rng.integers(1, 5))
    spread = rng.choice([1.0, 1.0, 3.0])  # sometimes coordinates outside the box
    dtype = rng.choice([np.float32, np.float64])
    strict = bool(rng.random() < 0.5)
    pos = [((rng.random((n, n_frames, 3)) - (0.5 if spread > 1 else 0.0)) * spread
            * np.asarray(box)).astype(np.float32).astype(dtype) for n in sizes]
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0).astype(np.float32), sizes, box, cutoff, nbins,
                            strict=strict)
    for literal in (False, True):
        eng = gpu.RdfEngine(sizes, box, cutoff, nbins, strict_reference_masks=strict,
                            literal_kernel=literal)
        if rng.random() < 0.5:
            eng.accumulate(pos)
        else:  # frame by frame, through a frame list, from device memory
            dev = [gpu.DeviceArray.from_numpy(p) for p in pos]
            for f in range(n_frames):
                eng.accumulate(dev, frames=np.array([f], dtype=np.int64))
        got = eng.read()
        eng.close()
        np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("seed", range(40))
def test_acf_random_window_layouts(gpu, seed):
    rng = np.random.default_rng(2000 + seed)
    n_series = int(rng.choice([1, 1, 1, 2, 5, 9, 40]))
    k = int(rng.choice([1, 3, 3, 3, 6]))
    data_range = int(rng.choice([4, 9, 64, 65, 130, 300]))
    ct = int(rng.choice([1, 1, 1, 2, 5, max(1, data_range // 3)]))
    n_frames = data_range + ct * int(rng.integers(5, 400)) + int(rng.integers(0, 7))
    dtype = rng.choice([np.float32, np.float64])
    x = (np.cumsum(rng.normal(size=(n_series, n_frames, k)), axis=1) * 0.03
         + rng.normal(size=(n_series, n_frames, k)) + rng.choice([0.0, 5.0])).astype(dtype)
    subsample = rng.random() < 0.25
    tau = (np.unique(np.concatenate([[0, data_range - 1],
                                     rng.integers(0, data_range, size=max(2, data_range // 3))]))
           if subsample else np.arange(data_range))
    n_win = (n_frames - data_range) // ct
    begin = int(rng.integers(0, max(1, n_win // 3)))
    end = int(rng.integers(begin + 1, n_win + 1))
    time = np.cumsum(np.concatenate([[0.0], 0.01 + 0.01 * rng.random(len(tau) - 1)]))
    scale, iscale = float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.1, 1.0))
    acf, mom = gpu.acf_windows(x, ct, begin, end, tau, scale=scale, integral_scale=iscale,
                               time=time)
    ref, per_window = c_oracle.acf_windows(x.astype(np.float64), ct, begin, end, tau, scale=scale,
                                           per_window=True)
    _close(acf, ref)
    sig = np.array([cumulative_trapezoid(iscale * r, x=time) for r in per_window])
    assert mom.n == end - begin
    _close(mom.mean, sig.mean(axis=0))
    _close(mom.std(), np.std(sig, axis=0))


@pytest.mark.parametrize("seed", range(30))
def test_msd_random_window_layouts(gpu, seed):
    rng = np.random.default_rng(3000 + seed)
    n_series = int(rng.choice([1, 3, 8, 9, 33, 70]))
    k = int(rng.choice([1, 3, 3, 3]))
    data_range = int(rng.choice([2, 15, 16, 17, 100, 129, 260]))
    ct = int(rng.choice([1, 1, 1, 3]))
    n_frames = data_range + ct * int(rng.integers(3, 300)) + int(rng.integers(0, 5))
    dtype = rng.choice([np.float32, np.float64])
    x = (np.cumsum(rng.normal(size=(n_series, n_frames, k)), axis=1) * 0.1
         + rng.uniform(-30, 30, size=(n_series, 1, k))).astype(dtype)
    tau = np.arange(data_range)
    if rng.random() < 0.2:
        tau = np.linspace(0, data_range - 1, max(2, data_range // 2), dtype=int)
    n_win = (n_frames - data_range) // ct
    begin = int(rng.integers(0, max(1, n_win // 2)))
    end = int(rng.integers(begin + 1, n_win + 1))
    scale = float(rng.uniform(0.2, 3.0))
    got = gpu.msd_windows(x, ct, begin, end, tau, scale=scale)
    ref = c_oracle.msd_windows(x.astype(np.float64), ct, begin, end, tau, scale=scale)
    _close(got, ref)


@pytest.mark.parametrize("seed", range(20))
def test_atom_sum_and_transformations_random_shapes(gpu, seed):
    from oracle import transformations as ort

    rng = np.random.default_rng(4000 + seed)
    n_atoms = int(rng.choice([1, 2, 31, 64, 100, 513]))
    n_frames = int(rng.choice([1, 2, 5, 63, 64, 257, 1025, 1500]))
    dtype = rng.choice([np.float32, np.float64])
    box = rng.uniform(5.0, 30.0, 3)
    # a random walk wrapped into the box: the unwrapping transformation has jumps to find
    walk = np.cumsum(rng.normal(scale=0.02 * box.min(), size=(n_atoms, n_frames, 3)), axis=1) \
        + rng.uniform(0, 1, (n_atoms, 1, 3)) * box
    wrapped = np.mod(walk, box).astype(dtype)
    w = rng.normal(size=n_atoms)
    np.testing.assert_allclose(
        gpu.weighted_atom_sum(wrapped, scalar_weight=0.7, weights=w),
        np.einsum("a,atc->tc", 0.7 * w, wrapped.astype(np.float64)), rtol=1e-12, atol=1e-10)
    got, _ = gpu.unwrap_coordinates(wrapped, box)
    ref, _ = ort.unwrap_coordinates(wrapped.astype(np.float64), box)
    np.testing.assert_array_equal(got, np.asarray(ref))
    rewrapped = gpu.map_coordinates(gpu.MAP_WRAP, got, box)
    np.testing.assert_array_equal(
        rewrapped, np.asarray(ort.wrap_coordinates(np.asarray(ref), box, center_box=False)))
    if n_frames >= 2:
        vel = gpu.map_coordinates(gpu.MAP_VELOCITY, got, [0.004, 0.0, 0.0])
        np.testing.assert_array_equal(
            vel, np.asarray(ort.velocity_from_positions(np.asarray(ref), 0.002, 2)))


@pytest.mark.parametrize("seed", range(16))
def test_adf_random_configurations(gpu, monkeypatch, seed):
    import itertools

    from oracle import adf as ora

    rng = np.random.default_rng(5000 + seed)
    n_species = int(rng.integers(1, 4))
    sizes = [int(rng.choice([1, 3, 20, 45, 80])) for _ in range(n_species)]
    box = [float(b) for b in rng.uniform(7.0, 14.0, 3)]
    cutoff = float(rng.uniform(1.5, 0.45 * min(box)))
    nbins = int(rng.choice([10, 64, 500]))
    power = int(rng.choice([0, 2, 4]))
    n_frames = int(rng.integers(1, 4))
    outside = rng.random() < 0.3  # coordinates several boxes away
    pos = [((rng.random((n, n_frames, 3)) - (2.0 if outside else 0.0)) * np.asarray(box)
            * (3.0 if outside else 1.0)).astype(np.float32) for n in sizes]
    # either pair search (the cell list needs three cells per axis, else it is not used)
    monkeypatch.setenv("MDX_ADF_CELL_MIN_ATOMS", "1" if rng.random() < 0.5 else "1000000")
    edges = np.histogram_bin_edges(np.zeros(1, dtype=np.float32), bins=nbins, range=(0.0, 3.15))
    hist, triples = gpu.adf_accumulate(pos, box, cutoff, edges, norm_power=power)
    ref = ora.adf_batch(pos, box, cutoff, nbins, power)
    combos = list(itertools.combinations_with_replacement(range(n_species), 3))
    for c, key in enumerate(combos):
        ref_hist, ref_count, ref_weight = ref[key]
        assert int(triples[c]) == ref_count, key
        if ref_count:
            np.testing.assert_allclose(hist[c].sum(), ref_weight, rtol=1e-6)
