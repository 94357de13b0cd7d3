"""CPU restatement (NumPy) of the reference's radial-distribution-function path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this module;
it is the checker used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.

PARITY UNPINNED: the reference (MDSuite 0.2.0) cannot be imported here
(TensorFlow is absent) and its own RDF tests compare against a downloaded JSON
at one decimal place, so no golden vector exists for this path.  The functions
below follow the reference source line by line, and restate the two TensorFlow
kernels involved from their published semantics:

* ``tf.math.rint``  -> round half to even (``np.rint``);
* ``tf.linalg.norm`` -> ``sqrt(reduce_sum(x * x, -1))`` as separate float32 ops,
  the three squares summed left to right;
* ``tf.histogram_fixed_width`` (CPU kernel, TensorFlow ``histogram_op.cc``)::

      step = double(range[1] - range[0]) / double(nbins)      # subtraction in float32
      bin  = int32(min(double(max(v, range[0]) - range[0]) / step, double(nbins - 1)))

All file:line citations are relative to the reference checkout.
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Sequence

import numpy as np

F32 = np.float32


def get_partial_triu_indices(n_atoms: int, m_atoms: int, idx: int) -> np.ndarray:
    """Rows ``idx .. idx+m_atoms-1`` of the strict upper triangle as ``[2, K]`` int32.

    mdsuite/utils/linalg.py:102-122: ``~band_part(ones(m, n), -1, idx)`` keeps the
    entries with ``column - row > idx``; ``tf.where`` lists them in row-major order.
    """
    rows = np.arange(m_atoms, dtype=np.int64)[:, None]
    cols = np.arange(n_atoms, dtype=np.int64)[None, :]
    mask = (cols - rows) > idx
    where = np.argwhere(mask).astype(np.int32)
    return where.T


def apply_minimum_image(r_ij: np.ndarray, box_array: np.ndarray) -> np.ndarray:
    """mdsuite/utils/linalg.py:84-99, every op in float32."""
    r_ij = np.asarray(r_ij, dtype=F32)
    box_array = np.asarray(box_array, dtype=F32)
    quotient = (r_ij / box_array).astype(F32)
    shift = (np.rint(quotient).astype(F32) * box_array).astype(F32)
    return (r_ij - shift).astype(F32)


def get_dij(indices: np.ndarray, positions_tensor: np.ndarray, atoms: np.ndarray,
            box_array: np.ndarray) -> np.ndarray:
    """mdsuite/calculators/radial_distribution_function.py:647-689.

    ``positions_tensor`` is ``[N, F, 3]`` float32, ``atoms`` the row block
    ``[m, F, 3]``; returns ``d_ij`` of shape ``[K, F]`` float32.
    """
    positions = np.take(positions_tensor, indices[1], axis=0)
    atoms_position = np.take(atoms, indices[0], axis=0)
    r_ij = (positions - atoms_position).astype(F32)
    if box_array is not None:
        r_ij = apply_minimum_image(r_ij, box_array)
    sq = (r_ij * r_ij).astype(F32)
    total = ((sq[..., 0] + sq[..., 1]).astype(F32) + sq[..., 2]).astype(F32)
    return np.sqrt(total).astype(F32)


def apply_system_cutoff(values: np.ndarray, cutoff) -> np.ndarray:
    """mdsuite/utils/linalg.py:125-136: keep ``values < cutoff`` (float32 compare)."""
    values = np.asarray(values, dtype=F32)
    return values[values < F32(cutoff)]


def histogram_fixed_width(values: np.ndarray, value_range, nbins: int) -> np.ndarray:
    """The CPU kernel behind ``tf.histogram_fixed_width`` (int32 counts)."""
    values = np.asarray(values, dtype=F32).ravel()
    lo = F32(value_range[0])
    hi = F32(value_range[1])
    step = np.float64(F32(hi - lo)) / np.float64(nbins)
    shifted = (np.maximum(values, lo) - lo).astype(F32)
    index = np.minimum(shifted.astype(np.float64) / step, np.float64(nbins - 1))
    index = index.astype(np.int32)  # truncation towards zero
    return np.bincount(index, minlength=nbins).astype(np.int32)


def bin_minibatch(start, stop, indices: np.ndarray, d_ij: np.ndarray, bin_range,
                  number_of_bins: int, cutoff) -> np.ndarray:
    """mdsuite/calculators/radial_distribution_function.py:616-645.

    ``indices`` is ``[K, 2]`` (local row, global column).  The comparisons are
    strict on both sides, as in the reference.
    """
    mask_1 = (indices[:, 0] > start[0]) & (indices[:, 0] < stop[0])
    mask_2 = (indices[:, 1] > start[1]) & (indices[:, 1] < stop[1])
    values_species = d_ij[mask_1 & mask_2]
    values = apply_system_cutoff(values_species, cutoff)
    return histogram_fixed_width(values, bin_range, number_of_bins)


def key_names(species: Sequence[str]) -> List[str]:
    """Pair keys ``"A_B"`` in combinations_with_replacement order (:272-275)."""
    index_list = range(len(species))
    return [f"{species[a]}_{species[b]}"
            for a, b in itertools.combinations_with_replacement(index_list, 2)]


def compute_species_values(indices: np.ndarray, start_batch: int, d_ij: np.ndarray,
                           particles_list: Sequence[int], species: Sequence[str],
                           bin_range, number_of_bins: int, cutoff) -> Dict[str, np.ndarray]:
    """mdsuite/calculators/radial_distribution_function.py:470-524."""
    rdf = {}
    indices_t = indices.T
    for a, b in itertools.combinations_with_replacement(range(len(species)), 2):
        name = f"{species[a]}_{species[b]}"
        start_ = np.array([sum(particles_list[:a]) - start_batch, sum(particles_list[:b])])
        stop_ = start_ + np.array([particles_list[a], particles_list[b]])
        rdf[name] = bin_minibatch(start_, stop_, indices_t, d_ij,
                                  np.asarray(bin_range, dtype=F32), number_of_bins,
                                  F32(cutoff))
    return rdf


def rdf_histograms(positions_by_species: Sequence[np.ndarray], species: Sequence[str],
                   box_array, cutoff, number_of_bins: int,
                   rdf_minibatch: int | None = None) -> Dict[str, np.ndarray]:
    """Histogram counts of one loaded batch, before the prefactor.

    Follows the batch body of ``run_calculator`` (:846-885): concatenate the
    species along the atom axis and cast to float32 (``_format_data`` :535-563),
    loop over row blocks of ``rdf_minibatch`` atoms (:867-872), build the partial
    triangle indices, distances, per-pair histograms, and add them up.
    ``positions_by_species[s]`` has shape ``[n_s, F, 3]``.  Returns int64 counts.
    """
    formatted = [np.asarray(p) for p in positions_by_species]
    positions_tensor = np.concatenate(formatted, axis=0).astype(F32)
    particles_list = [int(p.shape[0]) for p in formatted]
    n_atoms = positions_tensor.shape[0]
    if rdf_minibatch is None or rdf_minibatch <= 0:
        rdf_minibatch = n_atoms
    box = np.asarray(box_array, dtype=F32)
    bin_range = [0, cutoff]
    totals = {k: np.zeros(number_of_bins, dtype=np.int64) for k in key_names(species)}
    minibatch_start = 0
    for row0 in range(0, n_atoms, rdf_minibatch):
        atoms = positions_tensor[row0:row0 + rdf_minibatch]
        atoms_per_batch = atoms.shape[0]
        indices = get_partial_triu_indices(n_atoms, atoms_per_batch, minibatch_start)
        d_ij = get_dij(indices, positions_tensor, atoms, box)
        part = compute_species_values(indices, minibatch_start, d_ij, particles_list,
                                      species, bin_range, number_of_bins, cutoff)
        for k in totals:
            totals[k] += part[k].astype(np.int64)
        minibatch_start += atoms_per_batch
    return totals


# ----------------------------------------------------------------------------
# host-side pre/post-processing of the calculator


def default_arguments(box_array, n_configurations: int, number_of_configurations: int,
                      cutoff=None, number_of_bins=None, start: int = 0, stop=None):
    """``check_input`` + ``_initialize_rdf_parameters`` (:215-279)."""
    if stop is None:
        stop = n_configurations - 1
    if cutoff is None:
        cutoff = box_array[0] / 2 - 0.1
    if number_of_configurations == -1:
        number_of_configurations = n_configurations - 1
    if number_of_bins is None:
        number_of_bins = int(cutoff / 0.01)
    sample_configurations = np.linspace(start, stop, number_of_configurations, dtype=int)
    return dict(cutoff=cutoff, number_of_bins=number_of_bins, stop=stop,
                number_of_configurations=number_of_configurations,
                sample_configurations=sample_configurations)


def ideal_correction(box_array, cutoff, number_of_bins: int) -> np.ndarray:
    """``ideal_correction`` (:719-826), including its use of un-normalised radii."""
    def spherical(data):
        return 4 * np.pi * (data ** 2)

    def correction_1(data):
        return 2 * np.pi * data * (3 - 4 * data)

    def correction_2(data):
        arctan_1 = np.arctan(np.sqrt(4 * (data ** 2) - 2))
        arctan_2 = (8 * data * np.arctan((2 * data * (4 * (data ** 2) - 3))
                                         / (np.sqrt(4 * (data ** 2) - 2)
                                            * (4 * (data ** 2) + 1))))
        return 2 * data * (3 * np.pi - 12 * arctan_1 + arctan_2)

    bin_width = cutoff / number_of_bins
    bin_edges = np.linspace(0.0, cutoff, number_of_bins)
    lower_bound = box_array[0] / 2
    middle_bound = np.sqrt(2) * box_array[0] / 2
    first = bin_edges[bin_edges <= lower_bound]
    rest = bin_edges[~(bin_edges <= lower_bound)]
    if rest.size == 0:
        out = spherical(first)
    else:
        second = rest[rest < middle_bound]
        third = rest[~(rest < middle_bound)]
        if third.size == 0:
            out = np.concatenate((spherical(first), correction_1(second)))
        else:
            out = np.concatenate((spherical(first), correction_1(second),
                                  correction_2(third)))
    return out * bin_width


def radial_distribution_function(positions_by_species, species, n_particles, box_array,
                                 units_length: float, number_of_configurations: int = 500,
                                 cutoff=None, number_of_bins=None, start: int = 0,
                                 stop=None, minibatch: int = -1):
    """The whole calculator on in-memory data: returns ``{key: {"x": ..., "y": ...}}``.

    ``positions_by_species[s]`` is the stored dataset ``[n_s, n_configurations, 3]``
    (float32 on disk, mdsuite/database/simulation_database.py:491-497).  All sampled
    frames are histogrammed together; the split into load batches does not change
    integer counts.
    """
    n_cfg = positions_by_species[0].shape[1]
    args = default_arguments(box_array, n_cfg, number_of_configurations, cutoff,
                             number_of_bins, start, stop)
    cutoff = args["cutoff"]
    nbins = args["number_of_bins"]
    n_used = args["number_of_configurations"]
    rdf_minibatch = n_used if minibatch == -1 else minibatch
    sample = args["sample_configurations"]
    selected = [np.asarray(p)[:, sample] for p in positions_by_species]
    hist = rdf_histograms(selected, species, box_array, cutoff, nbins, rdf_minibatch)
    volume = float(np.prod(box_array))
    corr = ideal_correction(box_array, cutoff, nbins)
    out = {}
    with np.errstate(divide="ignore", invalid="ignore"):
        for key, counts in hist.items():
            a, b = key.split("_")
            scale = 2 if a == b else 1
            rho = n_particles[b] / volume
            denominator = n_used * rho * corr * n_particles[a]
            prefactor = scale / denominator
            y = counts.astype(float) * prefactor
            x = (units_length / 1e-9) * np.linspace(0.0, cutoff, nbins)
            out[key] = {"x": x.tolist(), "y": y.tolist()}
    return out, hist
