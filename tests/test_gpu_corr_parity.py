"""GPU parity of the atom-reduction, windowed ACF and MSD kernels against the oracle.

Tolerance for floating-point results (stated by the task's north star): 1e-6
relative to the largest magnitude of the compared series, because an
autocorrelation crosses zero and a per-element relative error is unbounded there.
The kernels sum in float64 directly while the reference uses an FFT, so the actual
differences are ~1e-13.
"""
import numpy as np
import pytest
from scipy.integrate import cumulative_trapezoid

from oracle import c_oracle
from oracle import correlation as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-6


def assert_close(got, ref, rtol=RTOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    scale = np.max(np.abs(ref)) if ref.size else 1.0
    np.testing.assert_allclose(got, ref, rtol=0.0, atol=rtol * max(scale, 1e-300))


# ------------------------------------------------------------------ reductions

@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(5, 7, 3), (33, 1000, 3), (129, 257, 3), (10, 64, 1)])
def test_ionic_current_matches_reference_transformation(gpu, dtype, shape):
    """Same check as the reference's test_ionic_current
    (CI/unit_tests/transformations/test_transformations.py:59-79), on the GPU."""
    rng = np.random.default_rng(7)
    out = None
    should_be = np.zeros((shape[1], shape[2]))
    species = []
    for _ in ("Na", "Cl"):
        vel = rng.random(shape).astype(dtype)
        charge = rng.random()
        species.append((vel, charge))
        should_be += np.sum(vel.astype(np.float64) * charge, axis=0)
    for i, (vel, charge) in enumerate(species):
        out = gpu.weighted_atom_sum(vel, scalar_weight=charge, out=out, accumulate=i > 0)
    np.testing.assert_allclose(out, should_be, rtol=1e-12, atol=1e-12)
    ref = orc.ionic_current([v for v, _ in species], [c for _, c in species])
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12)


def test_weighted_atom_sum_per_atom_weights_and_device_input(gpu):
    rng = np.random.default_rng(8)
    x = rng.normal(size=(77, 130, 3)).astype(np.float32)
    w = rng.normal(size=77)
    ref = np.einsum("a,atc->tc", w * 0.5, x.astype(np.float64))
    got = gpu.weighted_atom_sum(gpu.DeviceArray.from_numpy(x), scalar_weight=0.5, weights=w)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_series_can_stay_on_the_device(gpu):
    """Atom sums into a device array, float32 rounding by the scale map, correlation of the
    device-resident series: identical to the route through host arrays."""
    rng = np.random.default_rng(19)
    a = rng.normal(size=(50, 700, 3)).astype(np.float32)
    b = rng.normal(size=(30, 700, 3)).astype(np.float32)
    tau = np.arange(40)
    time = tau * 0.5
    host = gpu.weighted_atom_sum(a, scalar_weight=1.0)
    host = gpu.weighted_atom_sum(b, scalar_weight=-2.0, out=host, accumulate=True)
    series = host.astype(np.float32)[None]
    acf_h, mom_h = gpu.acf_windows(series, 1, 0, 660, tau, time=time)
    cur = gpu.DeviceArray((700, 3), np.float64)
    gpu.weighted_atom_sum(a, scalar_weight=1.0, out=cur)
    gpu.weighted_atom_sum(gpu.DeviceArray.from_numpy(b), scalar_weight=-2.0, out=cur,
                          accumulate=True)
    np.testing.assert_array_equal(cur.to_numpy(), host)
    rounded = gpu.DeviceArray((1, 700, 3), np.float32)
    gpu.map_coordinates(gpu.MAP_SCALE, cur.reshape((1, 700, 3)), [1.0, 1.0, 1.0], out=rounded)
    np.testing.assert_array_equal(rounded.to_numpy(), series)
    acf_d, mom_d = gpu.acf_windows(rounded, 1, 0, 660, tau, time=time)
    np.testing.assert_array_equal(acf_d, acf_h)
    np.testing.assert_array_equal(mom_d.m2, mom_h.m2)
    with pytest.raises(ValueError):
        cur.reshape((2, 700, 3))


def test_momentum_and_thermal_flux(gpu):
    rng = np.random.default_rng(9)
    n, t = 41, 97
    stress = rng.normal(size=(n, t, 6)).astype(np.float32)
    vel = rng.normal(size=(n, t, 3)).astype(np.float32)
    ke = rng.random((n, t, 1)).astype(np.float32)
    pe = rng.normal(size=(n, t, 1)).astype(np.float32)
    np.testing.assert_allclose(gpu.momentum_flux(stress), orc.momentum_flux([stress]),
                               rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(gpu.thermal_flux(stress, vel, ke, pe),
                               orc.thermal_flux(stress, vel, ke, pe), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("n,t,dtype", [
    (1, 2, np.float32), (7, 126, np.float32), (65, 128, np.float32), (33, 130, np.float32),
    (300, 1000, np.float32), (70, 4098, np.float32), (513, 640, np.float32),
    (40, 511, np.float32),    # odd frame count: the rows lose their alignment, generic kernels
    (40, 256, np.float64),    # float64 input: generic kernels
])
def test_flux_reductions_around_tile_edges(gpu, n, t, dtype):
    """MomentumFlux / ThermalFlux / IntegratedHeatCurrent sums
    (mdsuite/transformations/momentum_flux.py:45-55, thermal_flux.py:51-92,
    integrated_heat_current.py:48-60) against the oracle, at frame counts around the
    128-frame slices of the streaming kernels and atom counts around the chunking."""
    rng = np.random.default_rng(n * 10007 + t)
    stress = rng.normal(size=(n, t, 6)).astype(dtype)
    vel = rng.normal(size=(n, t, 3)).astype(dtype)
    ke = rng.random((n, t, 1)).astype(dtype)
    pe = rng.normal(size=(n, t, 1)).astype(dtype)
    tol = dict(rtol=0, atol=1e-12 * n)
    np.testing.assert_allclose(gpu.momentum_flux(stress), orc.momentum_flux([stress]), **tol)
    np.testing.assert_allclose(gpu.thermal_flux(stress, vel, ke, pe),
                               orc.thermal_flux(stress, vel, ke, pe), **tol)
    np.testing.assert_allclose(gpu.energy_weighted_sum(vel, ke, pe),
                               orc.integrated_heat_current([vel], [ke], [pe]), **tol)
    # accumulate over a second species, device-resident input
    dev = [gpu.DeviceArray.from_numpy(a) for a in (stress, vel, ke, pe)]
    out = gpu.thermal_flux(stress, vel, ke, pe)
    out = gpu.thermal_flux(*dev, out=out, accumulate=True)
    np.testing.assert_allclose(out, 2 * orc.thermal_flux(stress, vel, ke, pe),
                               rtol=0, atol=2e-12 * n)
    out = gpu.momentum_flux(dev[0], out=gpu.momentum_flux(stress), accumulate=True)
    np.testing.assert_allclose(out, 2 * orc.momentum_flux([stress]), rtol=0, atol=2e-12 * n)
    for d in dev:
        d.free()


# ------------------------------------------------------------------------- ACF

def _per_window_reference(x, ct, w0, w1, tau):
    """FFT estimator (the restated tfp.stats.auto_correlation) window by window."""
    out = []
    for w in range(w0, w1):
        win = np.take(x[:, w * ct: w * ct + tau[-1] + 1], tau, axis=1)
        out.append(np.sum(orc.auto_correlation(win, axis=1), axis=(0, 2)))
    return np.array(out)


@pytest.mark.parametrize("ct", [1, 7])
def test_system_series_acf_and_integral_moments(gpu, ct):
    rng = np.random.default_rng(10)
    T, D = 2048, 64
    x = np.cumsum(rng.normal(size=(1, T, 3)), axis=1) * 0.05 + rng.normal(size=(1, T, 3))
    x = x.astype(np.float32).astype(np.float64)          # series are stored as float32
    tau = np.arange(D)
    time = tau * 0.002
    n_win = int(np.clip((T - D) / ct, 1, None))
    acf, mom = gpu.acf_windows(x, ct, 0, n_win, tau, scale=1.0, time=time)
    ref_pw = _per_window_reference(x, ct, 0, n_win, tau)
    assert_close(acf, ref_pw.sum(axis=0))
    assert_close(acf, c_oracle.acf_windows(x, ct, 0, n_win, tau))
    sig = np.array([cumulative_trapezoid(r, x=time) for r in ref_pw])
    assert mom.n == n_win
    assert_close(mom.mean, sig.mean(axis=0))
    assert_close(mom.std(), np.std(sig, axis=0))


def test_acf_multi_series_float32_and_window_ranges(gpu):
    """Per-atom velocities: sums over atoms, split window ranges merged on the host."""
    rng = np.random.default_rng(11)
    S, T, D = 32, 1024, 128
    v = rng.normal(size=(S, T, 3)).astype(np.float32)
    tau = np.arange(D)
    time = tau * 0.1
    n_win = T - D
    ref_pw = _per_window_reference(v.astype(np.float64), 1, 0, n_win, tau)
    scale = 3.7
    acf_a, mom_a = gpu.acf_windows(v, 1, 0, 300, tau, scale=scale, integral_scale=1.0 / S,
                                   time=time)
    acf_b, mom_b = gpu.acf_windows(gpu.DeviceArray.from_numpy(v), 1, 300, n_win, tau,
                                   scale=scale, integral_scale=1.0 / S, time=time)
    assert_close(acf_a + acf_b, scale * ref_pw.sum(axis=0))
    mom = mom_a.merge(mom_b)
    sig = np.array([cumulative_trapezoid(scale * r / S, x=time) for r in ref_pw])
    assert_close(mom.mean, sig.mean(axis=0))
    assert_close(mom.std(), np.std(sig, axis=0))


def test_acf_subsampled_tau(gpu):
    """Non-contiguous tau values sub-sample the window before the correlation."""
    rng = np.random.default_rng(12)
    x = rng.normal(size=(2, 600, 3))
    tau = np.linspace(0, 99, 23, dtype=int)
    n_win = 40
    acf, _ = gpu.acf_windows(x, 5, 0, n_win, tau)
    ref = _per_window_reference(x, 5, 0, n_win, tau).sum(axis=0)
    assert_close(acf, ref)


def test_acf_direct_kernel_for_large_correlation_time(gpu):
    """Windows further apart than a quarter of their length share too few lag
    products to slide; they are evaluated one by one."""
    rng = np.random.default_rng(15)
    x = rng.normal(size=(3, 900, 3))
    tau = np.arange(64)
    time = tau * 0.5
    n_win = (900 - 64) // 20
    acf, mom = gpu.acf_windows(x, 20, 0, n_win, tau, scale=0.3, time=time)
    ref_pw = 0.3 * _per_window_reference(x, 20, 0, n_win, tau)
    assert_close(acf, ref_pw.sum(axis=0))
    sig = np.array([cumulative_trapezoid(r, x=time) for r in ref_pw])
    assert_close(mom.mean, sig.mean(axis=0))
    assert_close(mom.std(), np.std(sig, axis=0))


@pytest.mark.parametrize("S,T,D,k,ct,table_bytes", [
    (40, 1500, 300, 3, 1, None),        # two lag tiles, ragged edges
    (70, 900, 100, 1, 3, None),         # scalar series, two series chunks
    (24, 700, 65, 6, 2, None),          # six components (stress-like)
    (33, 1200, 128, 3, 1, 600_000),     # band table rebuilt for several window segments
])
def test_acf_many_series_band_table(gpu, monkeypatch, S, T, D, k, ct, table_bytes):
    """More than a handful of series: lag products are summed over series into a
    band table first, windows slide over that table."""
    if table_bytes is not None:
        monkeypatch.setenv("MDX_ACF_BAND_TABLE_BYTES", str(table_bytes))
    rng = np.random.default_rng(16)
    v = (rng.normal(size=(S, T, k)) + 0.3).astype(np.float32)
    tau = np.arange(D)
    time = tau * 0.01
    n_win = int(np.clip((T - D) / ct, 1, None))
    acf, mom = gpu.acf_windows(v, ct, 0, n_win, tau, scale=2.0, integral_scale=0.5, time=time)
    v64 = v.astype(np.float64)
    assert_close(acf, c_oracle.acf_windows(v64, ct, 0, n_win, tau, scale=2.0))
    ref_pw = 2.0 * _per_window_reference(v64, ct, 0, n_win, tau)
    assert_close(acf, ref_pw.sum(axis=0))
    sig = np.array([cumulative_trapezoid(0.5 * r, x=time) for r in ref_pw])
    assert mom.n == n_win
    assert_close(mom.mean, sig.mean(axis=0))
    assert_close(mom.std(), np.std(sig, axis=0))


def test_acf_sliding_sums_do_not_drift(gpu):
    """A long series with a large offset: the running window sums are restarted
    often enough that cancellation stays far below the tolerance."""
    rng = np.random.default_rng(17)
    T, D = 60_000, 200
    x = (rng.normal(size=(1, T, 3)) + 25.0).astype(np.float32)
    tau = np.arange(D)
    n_win = T - D
    acf, _ = gpu.acf_windows(x, 1, 0, n_win, tau)
    assert_close(acf, c_oracle.acf_windows(x.astype(np.float64), 1, 0, n_win, tau), rtol=1e-10)


@pytest.mark.parametrize("T,D,k,ct,begin,dtype", [
    (700, 5, 3, 1, 0, np.float32),       # fewer lags than one tile, ragged last strip
    (900, 130, 3, 1, 17, np.float64),    # four waves, windows start inside the series
    (2500, 257, 3, 1, 3, np.float32),    # eight waves, last tile one lag wide
    (3000, 700, 3, 1, 0, np.float64),    # two lag phases
    (1500, 96, 3, 5, 2, np.float32),     # several frames enter and leave per window
    (1200, 200, 1, 1, 0, np.float32),    # scalar series
    (1100, 150, 6, 2, 1, np.float64),    # six components
])
def test_acf_single_series_strip_kernel(gpu, T, D, k, ct, begin, dtype):
    """One series: windows in strips of 16, lags in tiles of 64, the integral's prefix
    sums joined across tiles and phases; every edge of that tiling."""
    rng = np.random.default_rng(18)
    x = (np.cumsum(rng.normal(size=(1, T, k)), axis=1) * 0.02
         + rng.normal(size=(1, T, k))).astype(dtype)
    tau = np.arange(D)
    time = np.cumsum(np.concatenate([[0.0], 0.01 + 0.005 * rng.random(D - 1)]))  # uneven
    end = (T - D) // ct
    acf, mom = gpu.acf_windows(x, ct, begin, end, tau, scale=1.5, integral_scale=0.25,
                               time=time)
    x64 = x.astype(np.float64)
    ref_pw = 1.5 * _per_window_reference(x64, ct, begin, end, tau)
    assert_close(acf, ref_pw.sum(axis=0))
    sig = np.array([cumulative_trapezoid(0.25 * r, x=time) for r in ref_pw])
    assert mom.n == end - begin
    assert_close(mom.mean, sig.mean(axis=0))
    assert_close(mom.std(), np.std(sig, axis=0))
    acf_only, _ = gpu.acf_windows(x, ct, begin, end, tau, scale=1.5)
    np.testing.assert_array_equal(acf_only, acf)


def test_acf_rejects_windows_beyond_the_batch(gpu):
    x = np.zeros((1, 100, 3))
    with pytest.raises(ValueError):
        gpu.acf_windows(x, 1, 0, 60, np.arange(50))


# ------------------------------------------------------------------------- MSD

@pytest.mark.parametrize("ct,dtype", [(1, np.float32), (3, np.float64)])
def test_msd_matches_oracle(gpu, ct, dtype):
    rng = np.random.default_rng(13)
    S, T, D = 50, 700, 100
    pos = np.cumsum(rng.normal(size=(S, T, 3)), axis=1).astype(dtype)
    tau = np.arange(D)
    n_win = int(np.clip((T - D) / ct, 1, None))
    got = gpu.msd_windows(pos, ct, 0, n_win, tau, scale=1.0)
    ref = np.zeros(D)
    p64 = pos.astype(np.float64)
    for w in range(n_win):
        ens = p64[:, w * ct: w * ct + D]
        ref += np.sum((ens - ens[:, None, 0]) ** 2, axis=(0, 2))
    assert_close(got, ref)
    assert_close(got, c_oracle.msd_windows(p64, ct, 0, n_win, tau))


def test_msd_subsampled_tau_and_scale(gpu):
    rng = np.random.default_rng(14)
    pos = np.cumsum(rng.normal(size=(1, 300, 3)), axis=1)
    tau = np.array([0, 3, 4, 10, 49])
    got = gpu.msd_windows(pos, 2, 5, 60, tau, scale=2.5)
    assert_close(got, c_oracle.msd_windows(pos, 2, 5, 60, tau, scale=2.5))


@pytest.mark.parametrize("S,T,D,k,begin,dtype", [
    (20, 900, 300, 3, 0, np.float32),      # wide lag tile, ragged window and lag edges
    (9, 500, 100, 3, 37, np.float64),      # narrow lag tile, window range with an offset
    (12, 400, 40, 2, 0, np.float32),       # two components
    (64, 300, 17, 1, 5, np.float64),       # scalar series, several series chunks
    (12, 503, 100, 3, 37, np.float32),     # float32: a four-frame load straddles the array end
    (10, 1001, 130, 3, 3, np.float32),     # the same with the wider lag tile
])
def test_msd_band_path_many_series(gpu, S, T, D, k, begin, dtype):
    """Consecutive origins, every lag and at least 8 series: squared displacements of a
    tile of origins x lags come from one staged image per series."""
    rng = np.random.default_rng(18)
    pos = (np.cumsum(rng.normal(size=(S, T, k)), axis=1) + 50.0).astype(dtype)
    tau = np.arange(D)
    end = T - D
    got = gpu.msd_windows(pos, 1, begin, end, tau, scale=0.25)
    assert_close(got, c_oracle.msd_windows(pos.astype(np.float64), 1, begin, end, tau,
                                           scale=0.25), rtol=1e-10)
    assert got[0] == 0.0


@pytest.mark.parametrize("dtype,offset", [(np.float32, 1.0e3), (np.float64, 1.0e6)])
@pytest.mark.parametrize("D", [100, 120, 300])
def test_msd_three_components_small_lags_far_from_origin(gpu, dtype, offset, D):
    """Three-component series take the matrix-core kernel, which expands the squared
    difference.  Slow walkers far from the origin (|x|^2 / msd(1) = 1e9 ... 1e15) must
    keep every lag to 1e-9 of its own value, not only of the largest lag."""
    rng = np.random.default_rng(23)
    S, T = 40, 1500
    pos = (np.cumsum(0.01 * rng.normal(size=(S, T, 3)), axis=1) + offset).astype(dtype)
    tau = np.arange(D)
    got = gpu.msd_windows(pos, 1, 3, T - D, tau)
    ref = c_oracle.msd_windows(pos.astype(np.float64), 1, 3, T - D, tau)
    assert got[0] == 0.0 and ref[0] == 0.0
    np.testing.assert_allclose(got[1:], ref[1:], rtol=1e-9, atol=0.0)


# ------------------------------------------------- Kinaci integrated heat current

@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n_atoms,n_frames", [(1, 1), (5, 7), (33, 1000), (130, 257), (7, 2049),
                                              (5, 4), (33, 132), (130, 256), (70, 2052),
                                              (300, 128)])
def test_kinaci_heat_current_matches_oracle(gpu, dtype, n_atoms, n_frames):
    """Running integral of force.velocity per atom in front of the atom reduction: float32
    arrays whose frame count is a multiple of four take the streaming form (slices of 128
    frames, the integral split at the slice's start), everything else the scan over blocks
    of 256 frames; sizes around the slice, block and chunk edges."""
    rng = np.random.default_rng(31)
    pos = np.cumsum(rng.normal(size=(n_atoms, n_frames, 3)), axis=1).astype(dtype)
    vel = rng.normal(size=(n_atoms, n_frames, 3)).astype(dtype)
    force = rng.normal(size=(n_atoms, n_frames, 3)).astype(dtype)
    pe = -rng.random((n_atoms, n_frames, 1)).astype(dtype)
    got, carry = gpu.kinaci_heat_current(pos, vel, force, pe, dt=0.002 * 5)
    ref, ref_carry = orc.kinaci_heat_current(pos, vel, force, pe, 0.002, 5)
    assert_close(got, ref, rtol=1e-12)
    assert_close(carry, ref_carry, rtol=1e-12)


def test_kinaci_heat_current_stream_and_scan_agree(gpu, monkeypatch):
    rng = np.random.default_rng(33)
    n_atoms, n_frames = 257, 1300
    pos = np.cumsum(rng.normal(size=(n_atoms, n_frames, 3)), axis=1).astype(np.float32)
    vel = rng.normal(size=(n_atoms, n_frames, 3)).astype(np.float32)
    force = rng.normal(size=(n_atoms, n_frames, 3)).astype(np.float32)
    pe = -rng.random((n_atoms, n_frames, 1)).astype(np.float32)
    start = rng.normal(size=n_atoms)
    streamed, carry_s = gpu.kinaci_heat_current(pos, vel, force, pe, dt=0.01, carry=start)
    monkeypatch.setenv("MDX_KINACI_SCAN", "1")
    scanned, carry_b = gpu.kinaci_heat_current(pos, vel, force, pe, dt=0.01, carry=start)
    assert_close(streamed, scanned, rtol=1e-13)
    assert_close(carry_s, carry_b, rtol=1e-13)
    ref, ref_carry = orc.kinaci_heat_current(pos, vel, force, pe, 0.01, 1,
                                             carry=start)
    assert_close(streamed, ref, rtol=1e-12)
    assert_close(carry_s, ref_carry, rtol=1e-12)


def test_kinaci_heat_current_carries_the_integral_across_batches(gpu):
    rng = np.random.default_rng(32)
    n_atoms, n_frames = 40, 900
    pos = np.cumsum(rng.normal(size=(n_atoms, n_frames, 3)), axis=1).astype(np.float32)
    vel = rng.normal(size=(n_atoms, n_frames, 3)).astype(np.float32)
    force = rng.normal(size=(n_atoms, n_frames, 3)).astype(np.float32)
    pe = -rng.random((n_atoms, n_frames, 1)).astype(np.float32)
    whole, whole_carry = orc.kinaci_heat_current(pos, vel, force, pe, 0.5, 2)
    parts, carry, ref_carry = [], None, None
    for lo, hi in ((0, 300), (300, 301), (301, 900)):
        cut = [np.ascontiguousarray(a[:, lo:hi]) for a in (pos, vel, force, pe)]
        part, carry = gpu.kinaci_heat_current(*cut, dt=1.0, carry=carry)
        ref_part, ref_carry = orc.kinaci_heat_current(*cut, 0.5, 2, carry=ref_carry)
        assert_close(part, ref_part, rtol=1e-12)
        parts.append(part)
    assert_close(np.concatenate(parts), whole, rtol=1e-12)
    assert_close(carry, whole_carry, rtol=1e-12)
    # a second species is added to the first one's frames
    out, _ = gpu.kinaci_heat_current(pos, vel, force, pe, dt=1.0)
    both, _ = gpu.kinaci_heat_current(pos[:9], vel[:9], force[:9], pe[:9], dt=1.0, out=out,
                                      accumulate=True)
    assert_close(both, whole + orc.kinaci_heat_current(pos[:9], vel[:9], force[:9], pe[:9],
                                                       0.5, 2)[0], rtol=1e-12)


@pytest.mark.parametrize("T,D,begin", [(1001, 128, 3), (777, 40, 1), (2050, 300, 5)])
def test_acf_band_float32_ragged_end(gpu, T, D, begin):
    """Three-component float32 series whose frame count, first window and lag range leave
    the last staged frames of a tile half inside the array: the staging reads four frames
    per lane and must take the part past the end as absent."""
    rng = np.random.default_rng(41)
    v = (rng.normal(size=(16, T, 3)) + 0.2).astype(np.float32)
    tau = np.arange(D)
    end = T - D + 1
    acf, _ = gpu.acf_windows(v, 1, begin, end, tau)
    assert_close(acf, c_oracle.acf_windows(v.astype(np.float64), 1, begin, end, tau), rtol=1e-10)
