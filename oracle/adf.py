"""TEST INFRASTRUCTURE - CPU restatement of the reference's angular distribution
function in NumPy, operation by operation and in the reference's precisions (float32
vectors, float16 cutoff test, float64 histogram).  Only tests may import this module;
the product never does.

PARITY UNPINNED: the reference's only ADF check compares with a downloaded JSON
(CI/integration_tests/calculators/test_angular_distribution_function.py), and
tf.math.acos / tf.pow are restated by their NumPy float32 counterparts.

Follows mdsuite/calculators/angular_distribution_function.py:385-475,
mdsuite/utils/neighbour_list.py:54-177 and mdsuite/utils/linalg.py:30-81.
"""
import itertools

import numpy as np


def neighbour_vectors(positions: np.ndarray, cell) -> np.ndarray:
    """``r_ij_mat`` of shape (T, N, N, 3), float32: minimum-image ``pos_i - pos_j`` for
    i < j and its negative for i > j (get_neighbour_list + scatter + ``-= transpose``,
    angular_distribution_function.py:340-366, neighbour_list.py:81-92)."""
    pos = np.asarray(positions, dtype=np.float32)          # (T, N, 3)
    n_atoms = pos.shape[1]
    iu, ju = np.triu_indices(n_atoms, k=1)
    r = pos[:, iu] - pos[:, ju]
    cell = np.asarray(cell, dtype=np.float32)
    r = r - np.rint(r / cell) * cell
    mat = np.zeros((pos.shape[0], n_atoms, n_atoms, 3), dtype=np.float32)
    mat[:, iu, ju] = r
    mat = mat - np.transpose(mat, (0, 2, 1, 3))
    return mat


def get_triplets(full_r_ij: np.ndarray, r_cut: float) -> np.ndarray:
    """All (t, i, j, k) with both distances inside the cutoff, tested in float16 as the
    reference does (neighbour_list.py:105-177); a zero distance counts as ``r_cut``."""
    r_ij = np.sqrt(np.sum(full_r_ij * full_r_ij, axis=-1, dtype=np.float32))
    r_ij = r_ij.astype(np.float16)
    cut = np.float16(r_cut)
    r_ij = np.where(r_ij == 0, cut, r_ij)
    inside = r_ij < cut                                    # (T, N, N)
    n_atoms = inside.shape[1]
    out = []
    for shift in range(1, n_atoms):
        rolled = np.roll(inside, -shift, axis=2)           # [t, i, j] -> inside[t, i, j+shift]
        t, i, j = np.nonzero(inside & rolled)
        k = (j + shift) % n_atoms
        out.append(np.stack([t, i, j, k], axis=1))
    return np.concatenate(out, axis=0) if out else np.zeros((0, 4), dtype=np.int64)


def get_angles(r_ij_mat: np.ndarray, triples: np.ndarray):
    """(angles, |r_ij| * |r_ik|) in float32 (linalg.py:30-81)."""
    r_ij = r_ij_mat[triples[:, 0], triples[:, 1], triples[:, 2]]
    r_ik = r_ij_mat[triples[:, 0], triples[:, 1], triples[:, 3]]
    n_ij = np.sqrt(np.sum(r_ij * r_ij, axis=-1, dtype=np.float32))
    n_ik = np.sqrt(np.sum(r_ik * r_ik, axis=-1, dtype=np.float32))
    u_ij = r_ij / n_ij[:, None]
    u_ik = r_ik / n_ik[:, None]
    dot = (u_ij[:, 0] * u_ik[:, 0] + u_ij[:, 1] * u_ik[:, 1]) + u_ij[:, 2] * u_ik[:, 2]
    angles = np.arccos(np.clip(dot, np.float32(-1.0), np.float32(1.0)))
    return angles.astype(np.float32), (n_ij * n_ik).astype(np.float32)


def adf_batch(positions_by_species, cell, cutoff: float, number_of_bins: int,
              norm_power: int, bin_range=(0.0, 3.15)):
    """One batch of ``_build_histograms`` (:385-425): ``{name: (density histogram float32,
    triple count, weight sum)}`` for every species triple in combination order."""
    sizes = [p.shape[0] for p in positions_by_species]
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    pos = np.transpose(np.concatenate(positions_by_species, axis=0), (1, 0, 2))
    mat = neighbour_vectors(pos, cell)
    triples = get_triplets(mat, cutoff)
    out = {}
    idx = list(range(len(sizes)))
    for a, b, c in itertools.combinations_with_replacement(idx, 3):
        cond = ((triples[:, 1] >= bounds[a]) & (triples[:, 1] < bounds[a + 1])
                & (triples[:, 2] >= bounds[b]) & (triples[:, 2] < bounds[b + 1])
                & (triples[:, 3] >= bounds[c]) & (triples[:, 3] < bounds[c + 1]))
        sel = triples[cond]
        angle_vals, pre_factor = get_angles(mat, sel)
        weights = (np.float32(1.0) / pre_factor ** np.float32(norm_power)).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            hist, _ = np.histogram(angle_vals, bins=number_of_bins, range=bin_range,
                                   weights=weights, density=True)
        out[(a, b, c)] = (hist.astype(np.float32), len(sel),
                          float(np.sum(weights, dtype=np.float64)))
    return out
