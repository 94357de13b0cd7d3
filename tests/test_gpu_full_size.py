"""GPU checks at the sizes the benchmark configurations name.

Where the CPU oracle finishes in seconds the comparison stays direct (one frame of
the 100k-atom water box); beyond that the results are checked through properties
that do not depend on the size: a histogram total against a closed form, and the
window-summed autocorrelation against its re-ordering as one weighted sum over the
series (every lag product counted once per window that covers it).
"""
import numpy as np
import pytest

from oracle import c_oracle

pytestmark = pytest.mark.gpu


def test_rdf_100k_atom_water_box_single_frame_bit_exact(gpu):
    """33 334 O + 66 666 H, L = 100, cutoff 49.9, 4 990 bins: 5e9 pairs in one frame."""
    rng = np.random.default_rng(4)
    n_per_species = [33_334, 66_666]
    box = [100.0, 100.0, 100.0]
    cutoff, nbins = np.float32(49.9), 4990
    pos = [(rng.random((n, 1, 3), dtype=np.float32) * np.float32(100.0)) for n in n_per_species]
    eng = gpu.RdfEngine(n_per_species, box, cutoff, nbins)
    eng.accumulate(pos)
    got = eng.read()
    eng.close()
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
    np.testing.assert_array_equal(got, ref)
    # strict reference masks drop the first atom of each species
    n_o, n_h = n_per_species[0] - 1, n_per_species[1] - 1
    assert got[0].sum() <= n_o * (n_o - 1) // 2
    assert got[1].sum() <= n_o * n_h
    assert got[2].sum() <= n_h * (n_h - 1) // 2


def test_rdf_10k_atoms_many_frames_total_is_additive(gpu):
    """Accumulating 64 frames in one call equals the sum of per-frame histograms, and
    sharding the tile pairs over four ranks changes nothing."""
    n_per_species = [5000, 5000]
    box = [65.0] * 3
    cutoff, nbins = np.float32(32.4), 3240
    n_frames = 64
    dev = [gpu.DeviceArray((n, n_frames, 3), np.float32) for n in n_per_species]
    for i, d in enumerate(dev):
        d.fill_uniform(seed=77 + i, lo=0.0, hi=65.0)
    whole = gpu.RdfEngine(n_per_species, box, cutoff, nbins)
    whole.accumulate(dev)
    total = whole.read()
    whole.close()
    parts = gpu.RdfEngine(n_per_species, box, cutoff, nbins)
    for f in range(0, n_frames, 16):
        for rank in range(4):
            parts.accumulate(dev, frames=np.arange(f, f + 16, dtype=np.int64),
                             shard_rank=rank, shard_count=4)
    np.testing.assert_array_equal(parts.read(), total)
    parts.close()
    for d in dev:
        d.free()


@pytest.mark.parametrize("data_range", [500, 5000])
def test_acf_million_frame_series_against_weighted_sum(gpu, data_range):
    """[1, 1e6, 3] AR(1) series, correlation_time 1.  With windows w = 0..n_win-1 of
    length D, sum_w sum_{n < D-m} P(w+n, m) = sum_t P(t, m) * c_m(t), where c_m(t)
    counts the windows whose lag-m range covers frame t."""
    rng = np.random.default_rng(5)
    T, D = 1_000_000, data_range
    phi = np.exp(-1.0 / 200.0)
    noise = rng.normal(size=(T, 3))
    from scipy.signal import lfilter
    x = lfilter([1.0], [1.0, -phi], noise, axis=0).astype(np.float32)[None]
    tau = np.arange(D)
    n_win = T - D
    acf, _ = gpu.acf_windows(x, 1, 0, n_win, tau)

    x64 = x[0].astype(np.float64)
    lags = np.unique(np.concatenate([np.arange(0, 8), np.linspace(8, D - 1, 25, dtype=int)]))
    t = np.arange(T)
    for m in lags:
        prod = np.einsum("td,td->t", x64[: T - m], x64[m:])
        tt = t[: T - m]
        # windows w with w <= t <= w + D-1-m, 0 <= w <= n_win-1
        lo = np.maximum(0, tt - (D - 1 - m))
        hi = np.minimum(n_win - 1, tt)
        count = np.clip(hi - lo + 1, 0, None)
        expected = float(np.dot(prod, count)) / (D - m)
        assert abs(acf[m] - expected) <= 1e-9 * abs(acf[0]), (m, acf[m], expected)


def test_large_host_arrays_take_the_page_locked_upload_path(gpu):
    """Host arrays of 96 MiB and more are sent in 64 MiB pieces through page-locked
    buffers filled by several threads: every byte must arrive, also when the size is not a
    multiple of anything, and a kernel fed from such an array must see the same data."""
    rng = np.random.default_rng(77)
    odd = rng.integers(0, 2 ** 31, size=(200 * 1024 * 1024 + 12_345) // 4, dtype=np.int32) \
        .view(np.float32)[:-1]
    dev = gpu.DeviceArray.from_numpy(odd)
    np.testing.assert_array_equal(dev.to_numpy().view(np.int32), odd.view(np.int32))
    dev.free()
    x = rng.random((300, 30_000, 3), dtype=np.float32)  # 108 MB
    got = gpu.weighted_atom_sum(x, scalar_weight=1.0)
    small = gpu.weighted_atom_sum(x[:, :100].copy(), scalar_weight=1.0)
    np.testing.assert_array_equal(got[:100], small)
    np.testing.assert_allclose(got, x.astype(np.float64).sum(axis=0), rtol=1e-12)


def test_frame_ranges_of_host_arrays_are_uploaded_row_by_row(gpu, monkeypatch):
    """A frame range ``stored[:, begin:end]`` of a larger host array (what the batch loader
    hands to the per-atom calculators) goes to the device row by row through the
    page-locked buffers, without a contiguous host copy: the bytes that arrive, and the
    results of the entry points that take such a view, equal those of a copy.  Row lengths
    that are no multiple of anything, more rows than one buffer holds, few long rows, and
    (threshold lowered) small arrays."""
    from lammps_analysis_amd import _lib
    rng = np.random.default_rng(78)
    lib = _lib.load()
    for n_rows, n_frames, begin, end in ((700, 20_000, 37, 19_950),    # 167 MB: several buffers
                                         (3, 9_000_000, 5, 8_999_999),   # rows longer than a buffer
                                         (513, 1_000, 1, 999)):
        stored = rng.integers(0, 2 ** 31, size=(n_rows, n_frames, 3), dtype=np.int32) \
            .view(np.float32)
        view = stored[:, begin:end]
        dev = gpu.DeviceArray(view.shape, np.float32)
        _lib.check(lib.mdx_memcpy_h2d_rows(dev.ptr, view.ctypes.data, n_rows,
                                           view.shape[1] * view.strides[1], view.strides[0]))
        np.testing.assert_array_equal(dev.to_numpy().view(np.int32),
                                      np.ascontiguousarray(view).view(np.int32))
        dev.free()
        del stored, view
    stored = np.cumsum(rng.normal(size=(400, 6_000, 3)), axis=1).astype(np.float32)  # 29 MB
    view = stored[:, 100:5_900]
    assert not view.flags.c_contiguous and gpu._is_frame_range(view)
    copy = np.ascontiguousarray(view)
    tau = np.arange(64)
    for arrays in ((view, copy),):
        a, b = arrays
        np.testing.assert_array_equal(gpu.weighted_atom_sum(a, scalar_weight=-1.5),
                                      gpu.weighted_atom_sum(b, scalar_weight=-1.5))
        got, mom = gpu.acf_windows(a, 1, 0, 5_000, tau)
        ref, ref_mom = gpu.acf_windows(b, 1, 0, 5_000, tau)
        np.testing.assert_array_equal(got, ref)
        np.testing.assert_array_equal(gpu.msd_windows(a, 1, 3, 5_000, tau),
                                      gpu.msd_windows(b, 1, 3, 5_000, tau))
        box = [50.0, 60.0, 70.0]
        np.testing.assert_array_equal(gpu.unwrap_coordinates(a, box)[0],
                                      gpu.unwrap_coordinates(b, box)[0])
    # entry points with several arrays: all of them frame ranges of stored arrays
    n, t = 300, 4_000
    stress, vel, force = (rng.normal(size=(n, t, k)).astype(np.float32) for k in (6, 3, 3))
    ke, pe = (rng.random((n, t, 1)).astype(np.float32) for _ in range(2))
    cut = np.s_[:, 8:3_992]
    views = [a[cut] for a in (stress, vel, force, ke, pe, stored[:n, :t])]
    copies = [np.ascontiguousarray(v) for v in views]
    assert all(gpu._is_frame_range(v, any_size=True) for v in views)
    for a, b in ((views, copies),):
        np.testing.assert_array_equal(gpu.thermal_flux(a[0], a[1], a[3], a[4]),
                                      gpu.thermal_flux(b[0], b[1], b[3], b[4]))
        np.testing.assert_array_equal(gpu.energy_weighted_sum(a[1], a[3], a[4]),
                                      gpu.energy_weighted_sum(b[1], b[3], b[4]))
        np.testing.assert_array_equal(gpu.kinaci_heat_current(a[5], a[1], a[2], a[4], 0.01)[0],
                                      gpu.kinaci_heat_current(b[5], b[1], b[2], b[4], 0.01)[0])
    # one contiguous array among them: everything stays on the host path
    np.testing.assert_array_equal(gpu.energy_weighted_sum(copies[1], views[3], views[4]),
                                  gpu.energy_weighted_sum(copies[1], copies[3], copies[4]))
    monkeypatch.setattr(gpu, "_ROWS_UPLOAD_MIN_BYTES", 0)
    tiny = stored[:7, 3:11]
    np.testing.assert_array_equal(gpu.weighted_atom_sum(tiny),
                                  gpu.weighted_atom_sum(np.ascontiguousarray(tiny)))


@pytest.mark.parametrize("what,data_range", [("msd", 100), ("msd", 300), ("acf", 500)])
def test_band_kernels_agree_at_size(gpu, monkeypatch, what, data_range):
    """Per-atom MSD / VACF of 3 000 atoms x 20 000 frames through the three band kernels
    (4x4x4 matrix instruction, 16x16x4 matrix instruction, vector unit): different tilings
    and, for the MSD, the expanded square against the literal one.  The small-size tests
    hold each of them to the oracle; here they must agree where the oracle cannot go."""
    n_atoms, n_frames = 3000, 20000
    x = gpu.DeviceArray((n_atoms, n_frames, 3), np.float32)
    x.fill_langevin(11, 0.05, 1.0)
    tau = np.arange(data_range)

    def run():
        if what == "msd":
            return gpu.msd_windows(x, 1, 0, n_frames - data_range, tau)
        return gpu.acf_windows(x, 1, 0, n_frames - data_range, tau)[0]

    results = {}
    for name, env in (("mfma4", {}), ("mfma16", {"MDX_BAND_MFMA16": "1"}),
                      ("valu", {"MDX_MSD_BAND_VALU": "1", "MDX_ACF_BAND_VALU": "1"})):
        for key in ("MDX_BAND_MFMA16", "MDX_MSD_BAND_VALU", "MDX_ACF_BAND_VALU"):
            monkeypatch.delenv(key, raising=False)
        for key, value in env.items():
            monkeypatch.setenv(key, value)
        results[name] = run()
    x.free()
    scale = np.max(np.abs(results["valu"]))
    for name in ("mfma4", "mfma16"):
        np.testing.assert_allclose(results[name], results["valu"], rtol=0.0, atol=1e-12 * scale)
    if what == "msd":
        assert results["mfma4"][0] == 0.0
