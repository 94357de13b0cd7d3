"""Guards of the default RDF kernel on the device: pairs built to sit on the worst places of
its error bound (the host-side model of tests/test_rdf_fast_bound.py, now through the real
kernels), and the flush that keeps its 32-bit LDS bins from wrapping.  The literal side is
radial_distribution_function.py:616-689 with linalg.py:84-99."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import rdf_fast_model as model

pytestmark = pytest.mark.gpu


def _three_kernels(gpu, n_per_species, box, cutoff, nbins, pos):
    out = {}
    for name, flags in (("fast", {}), ("table", {"table_kernel": True}),
                        ("literal", {"literal_kernel": True})):
        eng = gpu.RdfEngine(n_per_species, box, cutoff, nbins, strict_reference_masks=False,
                            **flags)
        eng.accumulate(pos)
        out[name] = eng.read()
        eng.close()
    return out


CLASSES = {
    # name: (box, cutoff, nbins)
    "fold_points": ((65.0, 65.0, 65.0), 32.4, 3240),
    "bin_edges": ((65.0, 65.0, 65.0), 32.4, 3240),
    "cutoff_edge": ((100.0, 100.0, 100.0), 49.9, 4990),
    "thin_box": ((10.0, 40.0, 160.0), 4.9, 490),
    "thin_box_long_cutoff": ((10.0, 40.0, 160.0), 60.0, 3000),
    "bins_at_lds_limit": ((65.0, 65.0, 65.0), 32.4, (1 << 14) - 3),
    "bins_power_of_two_minus_three": ((65.0, 65.0, 65.0), 32.4, (1 << 10) - 3),
}


@pytest.mark.parametrize("name", sorted(CLASSES))
def test_fast_kernel_on_pairs_built_against_its_bound(gpu, name):
    """Atoms come in pairs placed by the adversarial generators (coordinate differences at
    0, L/2, L; distances a few float32 steps either side of bin edges and of the cutoff).
    All pairs of a frame are evaluated, the built ones among them; the fast kernel, the
    table kernel, the literal kernel and the C oracle must agree count for count."""
    box, cutoff, nbins = CLASSES[name]
    cutoff = np.float32(cutoff)
    rng = np.random.default_rng(sum(map(ord, name)))
    n_pairs, frames = 600, 4
    per_frame = []
    for f in range(frames):
        if name == "fold_points":
            xi, xj = model.pairs_near_fold_points(box, n_pairs, rng)
        elif name == "cutoff_edge":
            xi, xj = model.pairs_near_bin_edges(box, cutoff, nbins, [nbins, nbins - 1],
                                                n_pairs, rng)
        else:
            edges = rng.integers(1, nbins + 1, 40)
            xi, xj = model.pairs_near_bin_edges(box, cutoff, nbins, edges, 2 * n_pairs // 40,
                                                rng)
        pick = rng.permutation(len(xi))[:n_pairs]
        assert len(pick) == n_pairs
        per_frame.append(np.concatenate([xi[pick], xj[pick]], axis=0))
    atoms = np.stack(per_frame, axis=1).astype(np.float32)   # [2 n_pairs][frames][3]
    n_per_species = [n_pairs, n_pairs]
    pos = [np.ascontiguousarray(atoms[:n_pairs]), np.ascontiguousarray(atoms[n_pairs:])]
    assert (atoms >= 0).all() and (atoms <= np.asarray(box, np.float32)).all()
    ref = c_oracle.rdf_hist(atoms, n_per_species, list(box), cutoff, nbins, strict=False)
    got = _three_kernels(gpu, n_per_species, list(box), cutoff, nbins, pos)
    for kernel, hist in got.items():
        np.testing.assert_array_equal(hist, ref, err_msg=f"{name}: {kernel}")


def test_lds_bins_are_flushed_before_they_wrap(gpu, monkeypatch):
    """One persistent block, every atom on one point: all pairs of all frames fall into bin
    0 of the block's 32-bit LDS histogram, 2 047 * 2 046 / 2 * 2 100 = 4.4e9 > 2^32 of
    them.  The count must equal the closed form."""
    monkeypatch.setenv("MDX_RDF_FAST_BLOCKS", "1")
    n_atoms, n_frames = 2048, 2100
    pos = gpu.DeviceArray((n_atoms, n_frames, 3), np.float32)
    pos.fill_uniform(seed=1, lo=3.0, hi=3.0)
    eng = gpu.RdfEngine([n_atoms], [13.0] * 3, np.float32(6.4), 640,
                        strict_reference_masks=True)
    eng.accumulate([pos])
    hist = eng.read()
    eng.close()
    pos.free()
    used = n_atoms - 1   # the strict masks drop the species' first atom
    expected = used * (used - 1) // 2 * n_frames
    assert expected > 2 ** 32
    assert int(hist[0, 0]) == expected
    assert int(hist.sum()) == expected
