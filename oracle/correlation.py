"""CPU restatement (NumPy/SciPy) of the reference's time-lag correlation path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this module;
it is the checker used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.

Pinning: the reference cannot be imported here (TensorFlow, TensorFlow-Probability
and JAX are absent), so these functions follow the reference source line by line
and restate ``tfp.stats.auto_correlation`` from its published algorithm.  They are
pinned by the reference's own offline-reproducible tests, restated in
``tests/test_oracle_reference_pins.py``: the exact batch/window integers of
CI/unit_tests/memory_manager/test_memory_manager.py:128-228, the transformation
identities of CI/unit_tests/transformations/test_transformations.py:59-79,124-144,
the analytic Langevin VACF and random-walk MSD of
CI/integration_tests/calculators/test_green_kubo_self_diffusion_coefficients.py:45-117
and test_einstein_diffusion_coefficients.py:45-99, and the helper-method checks of
CI/unit_tests/utils/test_calculator_helper_methods.py:42-133.  The Green-Kubo
viscosity / thermal-conductivity calculators have no enabled reference test:
PARITY UNPINNED for those two.

All file:line citations are relative to the reference checkout.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
from scipy.integrate import cumulative_trapezoid
from scipy.interpolate import UnivariateSpline
from scipy.optimize import curve_fit

# constants of mdsuite/utils/units.py:29-32
elementary_charge = 1.602176634e-19
boltzmann_constant = 1.380649e-23


# ----------------------------------------------------------------------------
# third-party arithmetic restated


def auto_correlation(x: np.ndarray, axis: int = -1) -> np.ndarray:
    """``tfp.stats.auto_correlation(x, axis, normalize=False, center=False)``.

    Published algorithm (tensorflow_probability/python/stats/sample_stats.py):
    move ``axis`` last, zero-pad the length-L series to ``2**ceil(log2(2 L))``,
    complex FFT, multiply by the conjugate, inverse FFT, keep the real part of
    lags ``0..L-1`` and divide lag ``m`` by ``L - m``.
    """
    x = np.asarray(x, dtype=np.float64)
    x = np.moveaxis(x, axis, -1)
    length = x.shape[-1]
    padded = int(2 ** np.ceil(np.log2(2.0 * length)))
    spectrum = np.fft.fft(x.astype(np.complex128), n=padded, axis=-1)
    product = np.fft.ifft(spectrum * np.conj(spectrum), axis=-1).real
    product = product[..., :length]
    product = product / (length - np.arange(length, dtype=np.float64))
    return np.moveaxis(product, -1, axis)


# ----------------------------------------------------------------------------
# batching: which windows exist (part of the numerical contract)


def linear_scale_function(memory_usage, scale_factor=1):
    """mdsuite/utils/scale_functions.py:30-47."""
    return memory_usage * scale_factor


def quadratic_scale_function(memory_usage, inner_scale_factor=1, outer_scale_factor=1):
    """mdsuite/utils/scale_functions.py:70-89."""
    return outer_scale_factor * (memory_usage * inner_scale_factor) ** 2


def get_batch_size(n_bytes_list: Sequence[float], n_configs: int, memory: float,
                   memory_fraction: float, scale_function=linear_scale_function,
                   scale_parameters: dict | None = None, offset: int = 0):
    """``MemoryManager.get_batch_size`` (memory_management/memory_manager.py:179-219).

    ``n_bytes_list`` holds the stored size of every dataset on the data path.
    Returns ``(batch_size, number_of_batches, remainder)``.
    """
    scale_parameters = scale_parameters or {}
    per_configuration_memory = 0.0
    for n_bytes in n_bytes_list:
        per_configuration_memory += n_bytes / n_configs
    per_configuration_memory = scale_function(per_configuration_memory, **scale_parameters)
    maximum_loaded = int(np.clip((memory_fraction * memory) / per_configuration_memory,
                                 1, n_configs - offset))
    batch_size = maximum_loaded
    number_of_batches, remainder = divmod(n_configs - offset, batch_size)
    return batch_size, number_of_batches, remainder


def get_ensemble_loop(batch_size: int, data_range: int, correlation_time: int = 1) -> int:
    """``MemoryManager.get_ensemble_loop`` without the mini-batch branch (:342-372)."""
    final_window = batch_size - data_range
    return int(np.clip(final_window / correlation_time, 1, None))


def batch_slices(n_configs: int, batch_size: int, n_batches: int, remainder: int):
    """Frame ranges yielded by ``DataManager.batch_generator.generator``
    (database/data_manager.py:156-221): ``n_batches`` full batches plus one
    remainder batch if ``remainder > 0``."""
    out = []
    for batch in range(n_batches + int(remainder > 0)):
        start = batch * batch_size
        stop = start + batch_size
        if batch == n_batches:
            stop = start + remainder
        out.append((start, stop))
    return out


def window_starts(data_size: int, data_range: int, correlation_time: int):
    """Window origins of one batch: ``dictionary_generator``
    (database/data_manager.py:309-339).  The count is recomputed from the batch's own
    ``data_size`` and the last admissible origin is excluded by the truncation."""
    ensemble_loop = int(np.clip((data_size - data_range) / correlation_time, 1, None))
    return [w * correlation_time for w in range(ensemble_loop)]


def handle_tau_values(tau_values, data_range: int):
    """``TrajectoryCalculator._handle_tau_values``
    (calculators/trajectory_calculator.py:196-228): returns (tau, data_range)."""
    if isinstance(tau_values, (int, np.integer)):
        tau_values = np.linspace(0, data_range - 1, int(tau_values), dtype=int)
    if isinstance(tau_values, (list, np.ndarray)):
        tau_values = np.asarray(tau_values)
        data_range = int(tau_values[-1] + 1)
    if isinstance(tau_values, slice):
        tau_values = np.linspace(0, data_range - 1, data_range, dtype=int)[tau_values]
    return np.asarray(tau_values), data_range


# ----------------------------------------------------------------------------
# atom reductions (transformations)


def ionic_current(velocities_by_species: Sequence[np.ndarray],
                  charges: Sequence[float]) -> np.ndarray:
    """``IonicCurrent.transform_batch`` (transformations/ionic_current.py:48-58):
    ``sum_species reduce_sum(charge * vel, axis=0)`` in float64 -> ``[T, 3]``.
    The same body serves ``TranslationalDipoleMoment``
    (transformations/translational_dipole_moment.py:52-62)."""
    total = None
    for vel, charge in zip(velocities_by_species, charges):
        vel = np.asarray(vel, dtype=np.float64)
        term = np.sum(np.float64(charge) * vel, axis=0)
        total = term if total is None else total + term
    return total


def momentum_flux(stress_by_species: Sequence[np.ndarray]) -> np.ndarray:
    """``MomentumFlux.transform_batch`` (transformations/momentum_flux.py:45-55)."""
    total = None
    for stress in stress_by_species:
        stress = np.asarray(stress, dtype=np.float64)
        phi = np.stack([stress[:, :, 3], stress[:, :, 4], stress[:, :, 5]], axis=2)
        term = np.sum(phi, axis=0)
        total = term if total is None else total + term
    return total


def thermal_flux(stress, vel, ke, pe) -> np.ndarray:
    """``ThermalFlux.transform_batch`` for one species
    (transformations/thermal_flux.py:51-92)."""
    stress = np.asarray(stress, dtype=np.float64)
    vel = np.asarray(vel, dtype=np.float64)
    ke = np.asarray(ke, dtype=np.float64)
    pe = np.asarray(pe, dtype=np.float64)
    phi_x = stress[:, :, 0] * vel[:, :, 0] + stress[:, :, 3] * vel[:, :, 1] + stress[:, :, 4] * vel[:, :, 2]
    phi_y = stress[:, :, 3] * vel[:, :, 0] + stress[:, :, 1] * vel[:, :, 1] + stress[:, :, 5] * vel[:, :, 2]
    phi_z = stress[:, :, 4] * vel[:, :, 0] + stress[:, :, 5] * vel[:, :, 1] + stress[:, :, 2] * vel[:, :, 2]
    phi = np.dstack([phi_x, phi_y, phi_z])
    phi_sum_atoms = phi.sum(axis=0)
    energy = ke + pe
    energy_velocity_atoms = np.sum(energy * vel, axis=0)
    return energy_velocity_atoms - phi_sum_atoms


def store_as_float32(series: np.ndarray) -> np.ndarray:
    """Derived series are written to a float32 dataset and read back as float64
    (transformations/transformations.py:289-291,315 and
    database/simulation_database.py:491-497,626)."""
    return np.asarray(series, dtype=np.float32).astype(np.float64)


# ----------------------------------------------------------------------------
# calculators


def _batches(n_configs, n_bytes_list, memory, memory_fraction, scale_factor):
    batch_size, n_batches, remainder = get_batch_size(
        n_bytes_list, n_configs, memory, memory_fraction, linear_scale_function,
        {"scale_factor": scale_factor})
    return batch_size, n_batches, remainder


def green_kubo_ionic_conductivity(current: np.ndarray, time_step: float, sample_rate: int,
                                  units, temperature: float, volume: float,
                                  data_range: int = 500, correlation_time: int = 1,
                                  tau_values=np.s_[:], integration_range=None,
                                  batch_size: int | None = None) -> Dict[str, list]:
    """``GreenKuboIonicConductivity`` (calculators/green_kubo_ionic_conductivity.py:113-310)
    on a stored ``[1, T, 3]`` ionic-current series (already float32-rounded).

    System properties are sliced on axis 0 of the ``(1, T, 3)`` dataset
    (database/data_manager.py:204-205), so batch 0 holds the whole series; the
    window count is computed from that batch's ``data_size`` = ``batch_size``.
    """
    current = np.asarray(current, dtype=np.float64)
    if integration_range is None:
        integration_range = data_range - 1
    tau, data_range = handle_tau_values(tau_values, data_range)
    time = np.asarray(tau) * time_step * sample_rate
    n_frames = current.shape[1]
    if batch_size is None:
        batch_size = n_frames
    prefactor = ((elementary_charge ** 2) * (units.length ** 2)
                 / (3 * boltzmann_constant * temperature * volume * units.volume * units.time))
    acf_array = np.zeros((data_range,))
    sigmas = []
    count = 0
    for start in window_starts(batch_size, data_range, correlation_time):
        ensemble = current[:, start:start + data_range]
        ensemble = np.take(ensemble, tau, axis=1)
        jacf = auto_correlation(ensemble, axis=1)
        jacf = np.squeeze(np.sum(jacf, axis=-1), axis=0)
        sigmas.append(cumulative_trapezoid(jacf, x=time))
        acf_array += jacf
        count += 1
    acf_array /= count
    sigma = cumulative_trapezoid(acf_array, x=time)
    sigma_sem = np.std(sigmas, axis=0) / np.sqrt(len(sigmas))
    return {
        "ionic_conductivity": [prefactor * sigma[integration_range - 1]],
        "uncertainty": [prefactor * sigma_sem[integration_range - 1]],
        "time": time.tolist(),
        "acf": acf_array.tolist(),
        "integral": sigma.tolist(),
        "integral_uncertainty": sigma_sem.tolist(),
    }


def green_kubo_self_diffusion(velocities: np.ndarray, time_step: float, sample_rate: int,
                              units, data_range: int = 500, correlation_time: int = 1,
                              tau_values=np.s_[:], integration_range=None,
                              batches: Sequence[tuple] | None = None) -> Dict[str, list]:
    """``GreenKuboDiffusionCoefficients`` for one species
    (calculators/green_kubo_self_diffusion_coefficients.py:111-337).
    ``velocities`` is ``[N, T, 3]``; ``batches`` lists the (start, stop) frame ranges
    of the load batches (default: one batch)."""
    velocities = np.asarray(velocities, dtype=np.float64)
    if integration_range is None:
        integration_range = data_range - 1
    tau, data_range = handle_tau_values(tau_values, data_range)
    time = np.asarray(tau) * time_step * sample_rate * units.time
    n_frames = velocities.shape[1]
    if batches is None:
        batches = [(0, n_frames)]
    acf_array = np.zeros(len(tau))
    sigmas = []
    count = 0
    for start, stop in batches:
        batch = velocities[:, start:stop]
        for origin in window_starts(stop - start, data_range, correlation_time):
            ensemble = batch[:, origin:origin + data_range]
            if ensemble.shape[1] != data_range:
                continue
            vacf = (units.length ** 2 / units.time ** 2) * auto_correlation(ensemble, axis=1)
            count += vacf.shape[0]
            sigmas.append(cumulative_trapezoid(np.sum(np.mean(vacf, axis=0), -1), x=time))
            acf_array += np.sum(np.sum(vacf, axis=0), -1)
            count += 1
    acf_array = acf_array / count
    sigmas = np.array(sigmas)
    sigma = cumulative_trapezoid(acf_array, x=time)
    sigma_sem = np.std(sigmas, axis=0) / np.sqrt(len(sigmas))
    return {
        "diffusion_coefficient": [1 / 3 * sigma[integration_range - 1]],
        "uncertainty": [1 / 3 * sigma_sem[integration_range - 1]],
        "time": time.tolist(),
        "acf": acf_array.tolist(),
        "integral": sigma.tolist(),
        "integral_uncertainty": sigma_sem.tolist(),
    }


def fit_einstein_curve(x_data: np.ndarray, y_data: np.ndarray, fit_max_index: int):
    """mdsuite/utils/calculator_helper_methods.py:41-107."""
    popt = []
    pcov = []

    def func(x, m, a):
        return m * x + a

    spline_data = UnivariateSpline(x_data, y_data, s=0, k=4)
    derivatives = spline_data.derivative(n=2)(x_data)
    derivatives[abs(derivatives) < 1e-5] = 0
    start_index = np.argmin(abs(derivatives))
    gradients = []
    gradient_errors = []
    for i in range(start_index + 2, len(y_data)):
        popt_temp, pcov_temp = curve_fit(func, xdata=x_data[start_index:i],
                                         ydata=y_data[start_index:i])
        gradients.append(popt_temp[0])
        gradient_errors.append(np.sqrt(np.diag(pcov_temp))[0])
        if i == fit_max_index:
            popt = popt_temp
            pcov = pcov_temp
    return popt, pcov, gradients, gradient_errors


def einstein_msd(positions: np.ndarray, data_range: int = 100, correlation_time: int = 1,
                 tau_values=np.s_[:], batches: Sequence[tuple] | None = None):
    """Ensemble loop of ``EinsteinDiffusionCoefficients``
    (calculators/einstein_diffusion_coefficients.py:168-190,217-248): returns
    ``(msd_sum, count, tau)`` with the reference's count (``n_atoms + 1`` per window)."""
    positions = np.asarray(positions, dtype=np.float64)
    tau, data_range = handle_tau_values(tau_values, data_range)
    n_frames = positions.shape[1]
    if batches is None:
        batches = [(0, n_frames)]
    msd_array = np.zeros(len(tau))
    count = 0
    for start, stop in batches:
        batch = positions[:, start:stop]
        for origin in window_starts(stop - start, data_range, correlation_time):
            ensemble = batch[:, origin:origin + data_range]
            if ensemble.shape[1] != data_range:
                continue
            msd = (np.take(ensemble, tau, axis=1) - ensemble[:, None, 0]) ** 2
            count += msd.shape[0]
            msd_array += np.sum(np.sum(msd, axis=0), axis=-1)
            count += 1
    return msd_array, count, tau


def einstein_diffusion(positions: np.ndarray, time_step: float, sample_rate: int, units,
                       data_range: int = 100, correlation_time: int = 1,
                       tau_values=np.s_[:], fit_range: int = -1,
                       batches: Sequence[tuple] | None = None) -> Dict[str, object]:
    """``EinsteinDiffusionCoefficients`` for one species (:112-248)."""
    if fit_range == -1:
        fit_range = int(data_range - 1)
    msd_array, count, tau = einstein_msd(positions, data_range, correlation_time,
                                         tau_values, batches)
    time = np.asarray(tau) * time_step * sample_rate
    msd_array = msd_array / count
    msd_array = msd_array * units.length ** 2
    time = time * units.time
    fit_values, covariance, gradients, gradient_errors = fit_einstein_curve(
        x_data=time, y_data=msd_array, fit_max_index=fit_range)
    error = np.sqrt(np.diag(covariance))[0]
    return {
        "diffusion_coefficient": 1 / 6.0 * fit_values[0],
        "uncertainty": 1 / 6.0 * error,
        "gradient": fit_values[0],
        "intercept": fit_values[1],
        "time": time.tolist(),
        "msd": msd_array.tolist(),
        "gradients": (np.array(gradients) / 6).tolist(),
        "gradient_errors": (np.array(gradient_errors) / 6).tolist(),
    }


def einstein_helfand_ionic_conductivity(dipole: np.ndarray, time_step: float,
                                        sample_rate: int, units, temperature: float,
                                        volume: float, data_range: int = 100,
                                        correlation_time: int = 1, tau_values=np.s_[:],
                                        fit_range: int = -1, n_batches: int = 1,
                                        batch_size: int | None = None) -> Dict[str, object]:
    """``EinsteinHelfandIonicConductivity``
    (calculators/einstein_helfand_ionic_conductivity.py:95-236) on a stored
    ``[1, T, 3]`` translational-dipole series.  The average uses the memory
    manager's ``n_batches * ensemble_loop`` (:161-163)."""
    dipole = np.asarray(dipole, dtype=np.float64)
    if fit_range == -1:
        fit_range = int(data_range - 1)
    tau, data_range = handle_tau_values(tau_values, data_range)
    time = np.asarray(tau) * time_step * sample_rate
    n_frames = dipole.shape[1]
    if batch_size is None:
        batch_size = n_frames
    prefactor = ((units.length ** 2) * (elementary_charge ** 2)
                 / (units.time * volume * units.volume * temperature * boltzmann_constant))
    msd_array = np.zeros(len(tau))
    for origin in window_starts(batch_size, data_range, correlation_time):
        ensemble = dipole[:, origin:origin + data_range]
        msd = (np.take(ensemble, tau, axis=1) - ensemble[:, 0, :]) ** 2
        msd = prefactor * np.sum(msd, axis=2)
        msd_array += msd[0, :]
    ensemble_loop = get_ensemble_loop(batch_size, data_range, correlation_time)
    msd_array /= int(n_batches) * ensemble_loop
    fit_values, covariance, gradients, gradient_errors = fit_einstein_curve(
        x_data=time, y_data=msd_array, fit_max_index=fit_range)
    error = np.sqrt(np.diag(covariance))[0]
    return {
        "ionic_conductivity": 1 / 6 * fit_values[0],
        "uncertainty": 1 / 6 * error,
        "time": time.tolist(),
        "msd": msd_array.tolist(),
    }


def green_kubo_windowed_flux(series: np.ndarray, data_range: int, correlation_time: int,
                             time: np.ndarray, integration_range: int,
                             batch_size: int | None = None):
    """The *intended* ensemble loop of GreenKuboViscosity / GreenKuboThermalConductivity /
    GreenKuboViscosityFlux (calculators/green_kubo_viscosity.py:185-208,
    green_kubo_thermal_conductivity.py:190-213, green_kubo_viscosity_flux.py:183-205):
    ``jacf = data_range * sum_dims auto_correlation(window)`` along the TIME axis,
    summed over windows, and ``trapz(jacf[:integration_range])`` per window.

    As written the reference correlates along axis 0 of a ``(1, D, 3)`` window (length
    1), which is inconsistent with the stored layout and has no enabled test; this
    function restates the time-axis reading.  PARITY UNPINNED.
    Returns ``(jacf_sum, sigma_per_window)``.
    """
    series = np.asarray(series, dtype=np.float64)
    n_frames = series.shape[1]
    if batch_size is None:
        batch_size = n_frames
    jacf_sum = np.zeros(data_range)
    sigma = []
    for origin in window_starts(batch_size, data_range, correlation_time):
        ensemble = series[:, origin:origin + data_range]
        jacf = data_range * np.sum(auto_correlation(ensemble, axis=1), axis=-1)[0]
        jacf_sum += jacf
        trapezoid = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
        sigma.append(trapezoid(jacf[:integration_range], x=time[:integration_range]))
    return jacf_sum, np.array(sigma)


def green_kubo_flux_as_written(series: np.ndarray, data_range: int, correlation_time: int,
                               time: np.ndarray, integration_range: int,
                               batch_size: int | None = None):
    """The ensemble loop of GreenKuboViscosity / GreenKuboThermalConductivity /
    GreenKuboViscosityFlux exactly as the files have it
    (calculators/green_kubo_viscosity.py:185-224): ``auto_correlation`` along AXIS 0 of
    the ``(1, D, 3)`` window, ``jacf`` of shape ``(1, D)``, ``np.trapz(jacf[:ir],
    x=time[:ir])`` per window (which slices the length-one axis), the sum of ``jacf`` over
    the windows.  Returns ``(jacf_sum (1, D), sigma (n_windows, 1))``; the calculators
    report ``prefactor * sigma[0]`` and ``prefactor * sigma[1]``.  PARITY UNPINNED (the
    reference's tests of these calculators are disabled)."""
    series = np.asarray(series, dtype=np.float64)
    n_frames = series.shape[1]
    if batch_size is None:
        batch_size = n_frames
    jacf_sum = np.zeros(data_range)
    sigma = []
    trapezoid = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
    for origin in window_starts(batch_size, data_range, correlation_time):
        ensemble = series[:, origin:origin + data_range]
        jacf = data_range * np.sum(auto_correlation(ensemble, axis=0), axis=-1)
        jacf_sum = jacf_sum + jacf
        sigma.append(trapezoid(jacf[:integration_range], x=time[:integration_range]))
    return jacf_sum, np.array(sigma)


# --------------------------------------------------------- distinct diffusion

def correlate(ds_a: np.ndarray, ds_b: np.ndarray) -> np.ndarray:
    """mdsuite/utils/calculator_helper_methods.py:110-148: ``jnp.correlate(a, b,
    "full")`` per dimension, mean over dimensions, lags ``0..T-1``; float64 here
    (the reference runs JAX at its float32 default)."""
    ds_a = np.asarray(ds_a, dtype=np.float64)
    ds_b = np.asarray(ds_b, dtype=np.float64)
    per_dim = [np.correlate(ds_a[:, d], ds_b[:, d], mode="full") for d in range(ds_a.shape[1])]
    acf = np.mean(per_dim, axis=0)
    return acf[int(len(acf) / 2):]


def msd_operation(ds_a: np.ndarray, ds_b: np.ndarray) -> np.ndarray:
    """mdsuite/utils/calculator_helper_methods.py:153-184."""
    ds_a = np.asarray(ds_a, dtype=np.float64)
    ds_b = np.asarray(ds_b, dtype=np.float64)
    return np.mean((ds_a - ds_a[0]) * (ds_b - ds_b[0]), axis=-1)


def map_over_particles(ds_a: np.ndarray, ds_b: np.ndarray, op) -> np.ndarray:
    """``_map_over_particles`` of both distinct calculators
    (green_kubo_distinct_diffusion_coefficients.py:192-238): mean over all (i, j) of
    ``op(a_i, b_j)``, literally as the double loop."""
    acc = None
    for a_i in ds_a:
        inner = np.mean([op(a_i, b_j) for b_j in ds_b], axis=0)
        acc = inner if acc is None else acc + inner
    return acc / len(ds_a)


def green_kubo_distinct_windows(vel_a: np.ndarray, vel_b: np.ndarray, same_species: bool,
                                data_range: int, correlation_time: int, time: np.ndarray):
    """Ensemble loop of GreenKuboDistinctDiffusionCoefficients over one batch
    (:240-290): returns (sum of window vacfs, list of trapezoid integrals)."""
    vel_a = np.asarray(vel_a, dtype=np.float64)
    vel_b = np.asarray(vel_b, dtype=np.float64)
    vacf_sum = np.zeros(data_range)
    sigma = []
    for start in window_starts(vel_a.shape[1], data_range, correlation_time):
        win_a = vel_a[:, start:start + data_range]
        win_b = vel_b[:, start:start + data_range]
        vacf = map_over_particles(win_a, win_b, correlate)
        if same_species:
            vacf = vacf - np.mean([correlate(x, y) for x, y in zip(win_a, win_b)], axis=0)
        vacf_sum += vacf
        sigma.append(np.trapezoid(vacf, x=time) if hasattr(np, "trapezoid")
                     else np.trapz(vacf, x=time))
    return vacf_sum, sigma


def einstein_distinct_windows(pos_a: np.ndarray, pos_b: np.ndarray, same_species: bool,
                              data_range: int, correlation_time: int) -> np.ndarray:
    """Ensemble loop of EinsteinDistinctDiffusionCoefficients over one batch
    (einstein_distinct_diffusion_coefficients.py:174-262): sum of the window msds."""
    pos_a = np.asarray(pos_a, dtype=np.float64)
    pos_b = np.asarray(pos_b, dtype=np.float64)
    msd_sum = np.zeros(data_range)
    for start in window_starts(pos_a.shape[1], data_range, correlation_time):
        win_a = pos_a[:, start:start + data_range]
        win_b = pos_b[:, start:start + data_range]
        msd = map_over_particles(win_a, win_b, msd_operation)
        if same_species:
            msd = msd - np.mean([msd_operation(x, y) for x, y in zip(win_a, win_b)], axis=0)
        msd_sum += msd
    return msd_sum


def integrated_heat_current(positions_by_species, ke_by_species, pe_by_species) -> np.ndarray:
    """mdsuite/transformations/integrated_heat_current.py:48-60: ``sum_species sum_atoms
    pos * (ke + pe)``, float64."""
    out = None
    for pos, ke, pe in zip(positions_by_species, ke_by_species, pe_by_species):
        term = np.sum(np.asarray(pos, np.float64)
                      * (np.asarray(ke, np.float64) + np.asarray(pe, np.float64)), axis=0)
        out = term if out is None else out + term
    return out


def kinaci_heat_current(pos, vel, force, pe, time_step, sample_rate, carry=None):
    """One species and one batch of mdsuite/transformations/
    kinaci_integrated_heat_current.py:53-87, float64: ``integrand = einsum('ijk,ijk->ij',
    force, vel)``, ``integral = carry[:, None] + cumsum(integrand, axis=1) * time_step *
    sample_rate``, ``r_k = einsum('ijk,ij->jk', pos, integral)``, ``r_p =
    einsum('ijk,ijm->jm', pe, pos)``; returns ``(r_k + r_p, integral[:, -1])``.

    The integral is kept per atom of the species.  The reference text adds the integral
    arrays of all species element by element (``tf.add_n(integrals)``), which only runs
    when every species has the same number of atoms and has no test ("someone who
    understands the transformation please provide a test", :90-91); that sum is not
    restated.  Parity unpinned."""
    pos = np.asarray(pos, np.float64)
    vel = np.asarray(vel, np.float64)
    force = np.asarray(force, np.float64)
    pe = np.asarray(pe, np.float64)
    integrand = np.einsum("ijk,ijk->ij", force, vel)
    integral = np.cumsum(integrand, axis=1) * time_step * sample_rate
    if carry is not None:
        integral = np.asarray(carry, np.float64)[:, None] + integral
    r_k = np.einsum("ijk,ij->jk", pos, integral)
    r_p = np.einsum("ijk,ijm->jm", pe, pos)
    return r_k + r_p, integral[:, -1].copy()
