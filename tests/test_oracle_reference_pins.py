"""The oracle against what the reference's own tests hold for this path (CPU only).

Restates, with NumPy inputs, the offline-reproducible checks of the reference:
  CI/unit_tests/memory_manager/test_memory_manager.py:128-228   (exact integers)
  CI/unit_tests/transformations/test_transformations.py:59-79,124-144
  CI/unit_tests/utils/test_calculator_helper_methods.py:42-133
  CI/integration_tests/calculators/test_green_kubo_self_diffusion_coefficients.py:45-117
  CI/integration_tests/calculators/test_einstein_diffusion_coefficients.py:45-99
The RDF has no offline check in the reference; its oracle is cross-checked between the
NumPy restatement (materialised index lists, gathers, masks) and the C restatement
(pair loops) and against hand-computable cases.
"""
import dataclasses

import numpy as np
import pytest

from oracle import c_oracle
from oracle import correlation as orc
from oracle import rdf as orc_rdf


@dataclasses.dataclass
class _Units:
    time: float = 1.0
    length: float = 1.0

    @property
    def volume(self):
        return self.length**3


# ------------------------------------------------------------------ batching

def test_batch_size_integers_of_the_reference():
    # data_size=500, rows=10, columns=10, fraction 0.5 of 50000 bytes, linear x10
    assert orc.get_batch_size([500], 10, 50000, 0.5, orc.linear_scale_function,
                              {"scale_factor": 10}) == (10, 1, 0)
    # data_size=500, rows=11, columns=13, fraction 1.0 of 50 bytes
    assert orc.get_batch_size([500], 13, 50, 1.0, orc.linear_scale_function,
                              {"scale_factor": 10}) == (1, 13, 0)
    assert orc.get_ensemble_loop(50, 10, 5) == 8
    assert orc.quadratic_scale_function(10, 2, 2) == 800


def test_window_origins_exclude_the_last_admissible_one():
    assert orc.window_starts(100, 10, 1) == list(range(90))
    assert orc.window_starts(100, 10, 7) == [7 * w for w in range(12)]
    assert orc.window_starts(10, 10, 1) == [0]      # clip(..., 1, None)
    assert orc.window_starts(5, 10, 1) == [0]


def test_tau_value_forms():
    tau, d = orc.handle_tau_values(np.s_[:], 6)
    assert tau.tolist() == [0, 1, 2, 3, 4, 5] and d == 6
    tau, d = orc.handle_tau_values(4, 10)
    assert tau.tolist() == [0, 3, 6, 9] and d == 10
    tau, d = orc.handle_tau_values([0, 2, 7], 100)
    assert tau.tolist() == [0, 2, 7] and d == 8


# ------------------------------------------------------------ transformations

def test_ionic_current_and_dipole_identity():
    rng = np.random.default_rng(0)
    vels, charges = [], []
    should_be = np.zeros((7, 3))
    for _ in ("Na", "Cl"):
        vel = rng.random((5, 7, 3))
        charge = rng.random()
        vels.append(vel)
        charges.append(charge)
        should_be += np.sum(vel * charge, axis=0)
    np.testing.assert_array_almost_equal(orc.ionic_current(vels, charges), should_be)
    np.testing.assert_array_almost_equal(
        c_oracle.weighted_atom_sum(vels[0].astype(np.float32), charges[0]),
        np.sum(vels[0].astype(np.float32).astype(np.float64) * charges[0], axis=0))


# --------------------------------------------------------------- helper methods

def test_fit_einstein_curve_slopes():
    x = np.linspace(0, 1000, 1000)
    popt, _, _, _ = orc.fit_einstein_curve(x, 5 * x + 3, 999)
    assert popt[0] == pytest.approx(5.0, 0.01)
    popt, _, _, _ = orc.fit_einstein_curve(x, np.exp(-0.05 * x) * x**2 + 5 * x + 3, 999)
    assert popt[0] == pytest.approx(5.0, 0.01)


# ------------------------------------------------------------- autocorrelation

def test_auto_correlation_equals_direct_unbiased_sum():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(3, 37, 2))
    got = orc.auto_correlation(x, axis=1)
    ref = np.zeros_like(got)
    for m in range(37):
        ref[:, m] = np.sum(x[:, :37 - m] * x[:, m:], axis=1) / (37 - m)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    tau = np.arange(37)
    acf = c_oracle.acf_windows(x, 1, 0, 1, tau)
    np.testing.assert_allclose(acf, ref.sum(axis=(0, 2)), rtol=1e-12, atol=1e-12)


def test_langevin_vacf_analytic():
    """Underdamped Langevin particles: VACF = v_th^2 exp(-t/tau), D known."""
    diff_coeff, kT, mass, time_step, n_part = 1.2345, 0.7, 1.5, 0.1, 100
    gamma = kT / diff_coeff
    relaxation_time = mass / gamma
    n_step = int(100 * relaxation_time / time_step)
    vacf_range = int(7 * relaxation_time / time_step)
    rng = np.random.default_rng(1)
    force = np.sqrt(2 * kT * gamma / time_step) * rng.normal(size=(n_step, n_part, 3))
    vel = np.zeros((n_step, n_part, 3))
    for i in range(1, n_step):
        vel[i] = vel[i - 1] + time_step * (force[i] - gamma * vel[i - 1]) / mass
    units = _Units(time=0.0056789, length=0.000007654)
    res = orc.green_kubo_self_diffusion(np.swapaxes(vel, 0, 1).astype(np.float32), time_step,
                                        1, units, data_range=vacf_range)
    t = time_step * np.arange(vacf_range) * units.time
    v_th = np.sqrt(3 * kT / mass) * units.length / units.time
    should = v_th**2 * np.exp(-t / (relaxation_time * units.time))
    np.testing.assert_allclose(res["time"], t, atol=1e-6)
    np.testing.assert_allclose(res["acf"], should, atol=0.03 * should[0])
    np.testing.assert_allclose(res["diffusion_coefficient"],
                               diff_coeff * units.length**2 / units.time, rtol=3e-2)


def test_random_walk_msd_analytic():
    rng = np.random.default_rng(864)
    diff_coeff, time_step, n_part, n_step, msd_range = 1.2345, 0.1, 200, 3000, 200
    vel = np.sqrt(2 * diff_coeff / time_step) * rng.normal(size=(n_step, n_part, 3))
    pos = time_step * np.cumsum(vel, axis=0)
    units = _Units(time=1.3, length=0.5)
    stored = np.swapaxes(pos, 0, 1).astype(np.float32)
    res = orc.einstein_diffusion(stored, time_step, 1, units, data_range=msd_range)
    t = time_step * np.arange(msd_range) * units.time
    d_should = diff_coeff * units.length**2 / units.time
    np.testing.assert_allclose(res["time"], t, atol=1e-5)
    np.testing.assert_allclose(res["msd"], 6 * d_should * t, rtol=9e-1)
    np.testing.assert_allclose(res["diffusion_coefficient"], d_should, rtol=2e-1)
    # C restatement against the NumPy one
    msd_sum, count, tau = orc.einstein_msd(stored, msd_range)
    n_win = n_step - msd_range
    np.testing.assert_allclose(c_oracle.msd_windows(stored, 1, 0, n_win, tau), msd_sum,
                               rtol=1e-11)
    assert count == n_win * (n_part + 1)


# ------------------------------------------------------------------------ RDF

def test_partial_triu_indices_cover_the_strict_upper_triangle():
    n = 7
    seen = []
    start = 0
    for m in (3, 3, 1):
        idx = orc_rdf.get_partial_triu_indices(n, m, start)
        seen += [(int(i) + start, int(j)) for i, j in idx.T]
        start += m
    assert sorted(seen) == [(i, j) for i in range(n) for j in range(i + 1, n)]


def test_histogram_fixed_width_rule():
    values = np.array([0.0, 0.0999, 0.1, 0.95, 1.0, 2.0, -1.0], dtype=np.float32)
    hist = orc_rdf.histogram_fixed_width(values, [0.0, 1.0], 10)
    # 0.0, -1.0 (clamped) -> bin 0; 0.0999 -> 0; 0.1 (float32 0.100000001) -> 1;
    # 0.95 -> 9; 1.0 and 2.0 -> clamped to the last bin
    assert hist.tolist() == [3, 1, 0, 0, 0, 0, 0, 0, 0, 3]


def test_strict_masks_drop_first_atom_of_each_species():
    # 3 + 3 atoms on a line, all within the cutoff
    pos = [np.array([[[0.0, 0, 0]], [[1.0, 0, 0]], [[2.0, 0, 0]]], dtype=np.float32),
           np.array([[[0.5, 0, 0]], [[1.5, 0, 0]], [[2.5, 0, 0]]], dtype=np.float32)]
    hist = orc_rdf.rdf_histograms(pos, ["A", "B"], [100.0] * 3, np.float32(10.0), 100)
    assert hist["A_A"].sum() == 1      # only (1,2) of species A
    assert hist["A_B"].sum() == 4      # {1,2} x {4,5}
    assert hist["B_B"].sum() == 1
    c = c_oracle.rdf_hist(np.concatenate(pos, 0), [3, 3], [100.0] * 3, 10.0, 100)
    assert c.sum(axis=1).tolist() == [1, 4, 1]
    c_all = c_oracle.rdf_hist(np.concatenate(pos, 0), [3, 3], [100.0] * 3, 10.0, 100,
                              strict=False)
    assert c_all.sum(axis=1).tolist() == [3, 9, 3]


@pytest.mark.parametrize("seed", [0, 1])
def test_numpy_and_c_rdf_restatements_agree(seed):
    rng = np.random.default_rng(seed)
    box = [9.0, 10.0, 11.0]
    pos = [(rng.random((n, 3, 3)) * np.asarray(box) * 3 - 9).astype(np.float32)
           for n in (17, 5, 23)]
    cutoff = np.float32(5.7)
    ref = orc_rdf.rdf_histograms(pos, ["a", "b", "c"], box, cutoff, 57, rdf_minibatch=11)
    got = c_oracle.rdf_hist(np.concatenate(pos, 0), [17, 5, 23], box, cutoff, 57)
    for i, key in enumerate(orc_rdf.key_names(["a", "b", "c"])):
        np.testing.assert_array_equal(got[i], ref[key].astype(np.uint64))


def test_rdf_ideal_gas_is_flat():
    """Uniform random positions: g(r) ~ 1 away from r = 0 (sanity of prefactor and
    ideal correction with the non-strict pair set being nearly the same)."""
    rng = np.random.default_rng(3)
    box = 20.0
    pos = [(rng.random((300, 20, 3)) * box).astype(np.float32)]
    out, _ = orc_rdf.radial_distribution_function(pos, ["Ar"], {"Ar": 300}, [box] * 3, 1e-10,
                                                  number_of_configurations=20, number_of_bins=50)
    y = np.asarray(out["Ar_Ar"]["y"])[10:]
    assert abs(np.mean(y) - 1.0) < 0.05


# ------------------------------------------------ coordinate transformations

def test_unwrap_coordinates_known_answer():
    """CI/unit_tests/transformations/test_transformations.py:144-188 of the reference."""
    from oracle import transformations as ort
    box_l = [1.1, 2.2, 3.3]
    pos = np.array([[[0.5, 0.1, 3.2]], [[0.6, 2.1, 0.9]], [[0.6, 2.1, 2.1]], [[0.6, 0.1, 0.1]]])
    pos = np.swapaxes(pos, 0, 1)
    carry = {"last_pos": np.array([[0.5, 0.1, 3.2]]), "last_image_box": np.array([[4.0, 0, 0]])}
    out, new_carry = ort.unwrap_coordinates(pos, box_l, carryover=carry)
    should_be = np.array([[[4 * 1.1 + 0.5, 0.1, 3.2]], [[4 * 1.1 + 0.6, -0.1, 4.2]],
                          [[4 * 1.1 + 0.6, -0.1, 5.4]], [[4 * 1.1 + 0.6, 0.1, 6.7]]])
    np.testing.assert_allclose(out, np.swapaxes(should_be, 0, 1), rtol=0, atol=1e-7)
    np.testing.assert_allclose(new_carry["last_pos"], [[0.6, 0.1, 0.1]])
    np.testing.assert_array_equal(new_carry["last_image_box"], [[4, 0, 2]])


def test_element_wise_coordinate_transformations():
    """test_scale_coordinates :103-121, test_unwrap_via_indices :191-213,
    test_velocity_from_positions :216-240, test_wrap_coordinates :243-262."""
    from oracle import transformations as ort
    rng = np.random.default_rng(5)
    pos = rng.random((5, 7, 3))
    box_l = np.array([1.1, 2.2, 3.3])
    np.testing.assert_array_equal(ort.scale_coordinates(pos, box_l), pos * box_l[None, None])
    box_im = rng.integers(-10, 10, size=(5, 7, 3)).astype(float)
    np.testing.assert_array_equal(ort.unwrap_via_indices(pos, box_im, box_l),
                                  pos + box_im * box_l[None, None])
    vels = (pos[:, 1:, :] - pos[:, :-1, :]) / (0.1 * 17)
    should_be = np.concatenate((vels, vels[:, -1:, :]), axis=1)
    np.testing.assert_array_equal(ort.velocity_from_positions(pos, 0.1, 17), should_be)
    wrapped = ort.wrap_coordinates(pos * 5 - 2, box_l, center_box=False)
    assert np.all(wrapped >= 0) and np.all(wrapped < box_l)
    centred = ort.wrap_coordinates(pos * 5 - 2, box_l, center_box=True)
    assert np.all(np.abs(centred) <= box_l / 2)


# ------------------------------------------------- angular distribution function

def test_adf_oracle_on_a_right_triangle():
    """No offline test of the reference pins the ADF; the restatement is at least held
    to geometry: a right isosceles triangle has one 90 degree and two 45 degree angles,
    each seen from both neighbour orders, weighted by 1 / (|r_ij| |r_ik|)^p."""
    from oracle import adf as ora
    pos = [np.array([[[1.0, 1.0, 1.0]], [[2.0, 1.0, 1.0]], [[1.0, 2.0, 1.0]]],
                    dtype=np.float32)]                       # [3 atoms, 1 frame, 3]
    nbins = 315                                              # 0.01 rad per bin
    out = ora.adf_batch(pos, [20.0, 20.0, 20.0], 3.0, nbins, 2)
    hist, count, weight = out[(0, 0, 0)]
    assert count == 6
    w90 = 1.0 / (1.0 * 1.0) ** 2
    w45 = 1.0 / (1.0 * np.sqrt(2.0)) ** 2
    np.testing.assert_allclose(weight, 2 * w90 + 4 * w45, rtol=1e-6)
    filled = np.nonzero(hist)[0]
    assert list(filled) == [int(np.pi / 4 / 0.01), int(np.pi / 2 / 0.01)]
    np.testing.assert_allclose(hist[filled[1]] / hist[filled[0]], (2 * w90) / (4 * w45),
                               rtol=1e-5)
    # the float16 cutoff test: 2.9995 rounds to 3.0 in float16 and is therefore outside,
    # 2.998 rounds to 2.998047 and stays inside
    for x, triples in ((2.9995, 0), (2.998, 2)):
        far = [np.array([[[0.0, 0.0, 0.0]], [[x, 0.0, 0.0]], [[0.0, 1.0, 0.0]]],
                        dtype=np.float32)]
        assert ora.adf_batch(far, [50.0] * 3, 3.0, 10, 0)[(0, 0, 0)][1] == triples


# ------------------------------------------------ RDF against lattice geometry

def _cubic_lattice(n=8, spacing=1.25):
    grid = np.arange(n, dtype=np.float32) * np.float32(spacing)
    pos = np.stack(np.meshgrid(grid, grid, grid, indexing="ij"), axis=-1).reshape(-1, 1, 3)
    return np.ascontiguousarray(pos), [n * spacing] * 3


def lattice_shell_counts(n=8, spacing=1.25, cutoff=4.9):
    """Pairs per squared lattice distance k = dx^2 + dy^2 + dz^2 (in units of the spacing)
    of a periodic simple cubic lattice, by enumeration of lattice vectors (every atom
    sees the same shells)."""
    shells = {}
    half = n // 2
    for dx in range(-half, half + 1):
        for dy in range(-half, half + 1):
            for dz in range(-half, half + 1):
                if (dx, dy, dz) == (0, 0, 0) or half in (abs(dx), abs(dy), abs(dz)):
                    continue                       # no image ambiguity below the cutoff
                k = dx * dx + dy * dy + dz * dz
                if spacing * np.sqrt(k) < cutoff:
                    shells[k] = shells.get(k, 0) + 1
    return {k: c * n**3 // 2 for k, c in shells.items()}


def lattice_histogram(nbins, cutoff, n=8, spacing=1.25):
    """Expected RDF counts of the lattice: every shell distance is sqrt of an exactly
    representable float32, binned by the histogram rule of the reference."""
    expected = np.zeros(nbins, dtype=np.uint64)
    step = float(np.float32(cutoff)) / nbins
    for k, count in lattice_shell_counts(n, spacing, float(cutoff)).items():
        d = np.sqrt(np.float32(k * spacing * spacing))
        expected[int(min(float(d) / step, nbins - 1))] += count
    return expected


def test_rdf_oracles_count_the_shells_of_a_cubic_lattice():
    """The RDF restatement has no reference fixture to lean on; a periodic simple cubic
    lattice gives exact pair counts per shell (6, 12, 8, 6, 24, ... neighbours)."""
    pos, box = _cubic_lattice()
    cutoff, nbins = np.float32(4.9), 490
    hist = c_oracle.rdf_hist(pos, [len(pos)], box, cutoff, nbins, strict=False)[0]
    shells = lattice_shell_counts()
    assert shells[1] == 512 * 6 // 2 and shells[2] == 512 * 12 // 2 and shells[3] == 512 * 8 // 2
    np.testing.assert_array_equal(hist, lattice_histogram(nbins, cutoff))
    # the literal NumPy restatement keeps the reference's strict masks, which drop the
    # first atom of the species: each shell loses that atom's own neighbours
    strict = orc_rdf.rdf_histograms([pos], ["Ar"], box, cutoff, nbins, rdf_minibatch=64)
    full = lattice_histogram(nbins, cutoff).astype(np.int64)
    np.testing.assert_array_equal(strict["Ar_Ar"].astype(np.int64), full - full // 256)


def cubic_lattice_adf_expectation(n_side, nbins, power, spacing=2.0):
    """Known answer for a periodic simple cubic lattice with the cutoff between the first and
    second shell: every atom has 6 neighbours, 30 ordered neighbour pairs, 24 of them at a
    right angle and 6 opposite; all vectors, unit vectors and cosines are exact in float32."""
    n_atoms = n_side ** 3
    edges = np.histogram_bin_edges(np.zeros(1, dtype=np.float32), bins=nbins, range=(0.0, 3.15))
    weight = 1.0 / (spacing * spacing) ** power
    expected = np.zeros(nbins)
    for angle, count in ((np.float32(np.arccos(np.float32(0.0))), 24),
                         (np.float32(np.arccos(np.float32(-1.0))), 6)):
        expected[np.searchsorted(edges, np.float64(angle), side="right") - 1] += \
            count * n_atoms * weight
    return edges, 30 * n_atoms, expected


def cubic_lattice_positions(n_side, n_frames, spacing=2.0):
    grid = np.arange(n_side, dtype=np.float32) * np.float32(spacing)
    pos = np.stack(np.meshgrid(grid, grid, grid, indexing="ij"), axis=-1).reshape(-1, 3)
    return np.repeat(pos[:, None, :], n_frames, axis=1).astype(np.float32)


def test_adf_oracle_counts_the_angles_of_a_cubic_lattice():
    from oracle import adf as ora

    n_side, nbins, power = 4, 63, 2
    pos = cubic_lattice_positions(n_side, 1)
    box = [n_side * 2.0] * 3
    edges, triples, expected = cubic_lattice_adf_expectation(n_side, nbins, power)
    hist, count, weight_sum = ora.adf_batch([pos], box, 2.4, nbins, power)[(0, 0, 0)]
    assert count == triples
    np.testing.assert_allclose(weight_sum, expected.sum(), rtol=1e-6)
    density = expected / np.diff(edges) / expected.sum()
    np.testing.assert_allclose(hist, density.astype(np.float32), rtol=1e-5)
