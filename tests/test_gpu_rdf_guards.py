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
    # the position field holds nbins + 2 + 64 dump words: the last count that fits 2^12 and
    # the first that needs 2^13
    "bins_dump_words_just_fit": ((65.0, 65.0, 65.0), 32.4, (1 << 12) - 66),
    "bins_dump_words_need_a_bit": ((65.0, 65.0, 65.0), 32.4, (1 << 12) - 65),
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


def _culled_and_not(gpu, monkeypatch, n_per_species, box, cutoff, nbins, pos, strict):
    """Histograms with the frames sorted in space for skipping, for leaving out folds, for
    both, and not sorted at all."""
    out = {}
    for name, cull, unfold in (("culled", "1", "1"), ("culled_every_fold", "1", "0"),
                               ("unfolded", "0", "1"), ("all_pairs", "0", "0")):
        monkeypatch.setenv("MDX_RDF_CULL", cull)
        monkeypatch.setenv("MDX_RDF_UNFOLD", unfold)
        eng = gpu.RdfEngine(n_per_species, box, cutoff, nbins, strict_reference_masks=strict)
        eng.accumulate(pos)
        out[name] = eng.read()
        eng.close()
    monkeypatch.delenv("MDX_RDF_CULL")
    monkeypatch.delenv("MDX_RDF_UNFOLD")
    return out


@pytest.mark.parametrize("case", ["uniform", "rectangular", "faces", "clustered", "shifted",
                                  "beyond_the_box", "one_sparse_species", "long_cutoff"])
def test_short_cutoff_path_skips_no_pair(gpu, monkeypatch, case):
    """Frames sorted in space, tile pairs out of reach skipped (the path short cutoffs take)
    against the same engine with the skipping off and against the C oracle: uniform and
    clustered atoms, a rectangular box, atoms exactly on the box faces, a trajectory that
    is not wrapped into [0, L), frames spread over several boxes (left to the table kernel),
    a species too small to be sorted into more than one tile, and a cutoff at which
    nothing can be skipped."""
    rng = np.random.default_rng(sum(map(ord, case)))
    n_per_species, frames = [2900, 1700], 5
    box = np.array([40.0, 40.0, 40.0], np.float32)
    cutoff, nbins, strict = np.float32(5.0), 500, True
    if case == "rectangular":
        box = np.array([24.0, 40.0, 72.0], np.float32)
    if case == "one_sparse_species":
        n_per_species = [4000, 37]
    if case == "long_cutoff":
        cutoff, nbins = np.float32(19.9), 1990
    pos = [(rng.random((n, frames, 3)) * box).astype(np.float32) for n in n_per_species]
    if case == "faces":
        for p in pos:
            p[::3, :, 0] = 0.0
            p[1::3, :, 0] = box[0]
            p[::5, :, 2] = np.nextafter(box[2], np.float32(0.0))
    elif case == "clustered":
        centres = (rng.random((12, 3)) * box).astype(np.float32)
        pos = [(centres[rng.integers(0, 12, n)][:, None, :]
                + rng.normal(scale=1.5, size=(n, frames, 3))).astype(np.float32) % box
               for n in n_per_species]
        strict = False
    elif case == "shifted":
        for p in pos:
            p += np.float32(-17.25)          # coordinates in [-17.25, L - 17.25)
    elif case == "beyond_the_box":
        pos[0][::9, 1] += np.float32(2.0) * box  # frame 1 spreads over three boxes
        pos[1][::4, 3] -= np.float32(0.3) * box  # frame 3 a little past a face
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, list(box), cutoff, nbins,
                            strict=strict)
    got = _culled_and_not(gpu, monkeypatch, n_per_species, list(box), cutoff, nbins, pos, strict)
    for name in ("all_pairs", "culled", "culled_every_fold", "unfolded"):
        np.testing.assert_array_equal(got[name], ref, err_msg=name)
    assert ref.sum() > 0


def test_short_cutoff_path_is_chosen_by_the_geometry(gpu, monkeypatch):
    """Without the override the library sorts and skips when the cutoff is short against
    the box and the species have tiles to skip, and only then; either way the counts are
    the oracle's.  20 000 atoms, cutoff 6 in a box of 60: most tile pairs are out of reach."""
    monkeypatch.delenv("MDX_RDF_CULL", raising=False)
    monkeypatch.delenv("MDX_RDF_UNFOLD", raising=False)
    rng = np.random.default_rng(11)
    n_per_species, box = [7000, 13000], [60.0, 60.0, 60.0]
    pos = [(rng.random((n, 2, 3)) * 60.0).astype(np.float32) for n in n_per_species]
    for cutoff, nbins in ((np.float32(6.0), 600), (np.float32(29.9), 2990)):
        eng = gpu.RdfEngine(n_per_species, box, cutoff, nbins)
        eng.accumulate(pos)
        got = eng.read()
        eng.close()
        ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
        np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("case", ["half_a_box_apart", "planes", "rectangular", "drifted",
                                  "small_species"])
def test_components_without_the_fold_match_the_oracle(gpu, monkeypatch, case):
    """Sorted frames, long cutoff: the pair loop leaves out the minimum-image fold of a
    component when the boxes of the two 64-atom groups say that no difference exceeds half
    a box length.  Cases around that limit: two sheets of atoms exactly and almost half a
    box apart (differences of exactly L/2, which the literal rule does not fold, next to
    differences one float above, which it does), atoms on a few planes (groups of zero
    extent), a rectangular box (the limit differs per component), a trajectory far from
    the origin, a species of less than one group beside a large one."""
    rng = np.random.default_rng(sum(map(ord, case)) + 5)
    n_per_species, frames = [4200, 3100], 3
    box = np.array([36.0, 36.0, 36.0], np.float32)
    cutoff, nbins = np.float32(17.9), 1790
    if case == "rectangular":
        box = np.array([20.0, 36.0, 90.0], np.float32)
        cutoff, nbins = np.float32(9.9), 990
    if case == "small_species":
        n_per_species = [6000, 41]
    pos = [(rng.random((n, frames, 3)) * box).astype(np.float32) for n in n_per_species]
    if case == "half_a_box_apart":
        half = np.float32(18.0)
        for p in pos:
            sheet = rng.integers(0, 4, p.shape[0])
            jitter = (rng.integers(-2, 3, p.shape[:2]) * np.float32(2.0 ** -20)).astype(np.float32)
            # x on two sheets exactly half a box apart, a few ulps around them
            p[:, :, 0] = np.where((sheet % 2 == 0)[:, None], np.float32(4.0), np.float32(4.0) + half) + jitter
            # y on two sheets a little less than half a box apart
            p[:, :, 1] = np.where((sheet // 2 == 0)[:, None], np.float32(1.0), np.float32(18.75))
    elif case == "planes":
        for p in pos:
            p[:, :, 2] = (rng.integers(0, 6, p.shape[:2]) * np.float32(6.0)).astype(np.float32)
            p[:, :, 0] = np.round(p[:, :, 0] * 2) / 2
    elif case == "drifted":
        for p in pos:
            p += np.array([1000.0, -250.5, 36.0], np.float32)
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, list(box), cutoff, nbins,
                            strict=True)
    got = _culled_and_not(gpu, monkeypatch, n_per_species, list(box), cutoff, nbins, pos, True)
    for name in ("all_pairs", "unfolded", "culled"):
        np.testing.assert_array_equal(got[name], ref, err_msg=name)
    assert ref.sum() > 0


@pytest.mark.parametrize("cutoff,nbins", [(np.float32(5.0), 500), (np.float32(19.9), 1990)])
def test_sorted_frames_with_the_tile_pairs_sharded(gpu, monkeypatch, cutoff, nbins):
    """Ranks that share the tile pairs of a frame sort it each for themselves, so the sorted
    order must be a function of the frame alone: three shards, each packing and sorting the
    same frames in a call of its own, add up to the oracle's histogram (short cutoff with
    skipping, long cutoff with folds left out)."""
    monkeypatch.setenv("MDX_RDF_CULL", "1" if cutoff < 10 else "0")
    monkeypatch.setenv("MDX_RDF_UNFOLD", "1")
    rng = np.random.default_rng(23)
    n_per_species, box, frames = [3300, 2100], [40.0, 40.0, 40.0], 4
    pos = [(rng.random((n, frames, 3)) * 40.0).astype(np.float32) for n in n_per_species]
    eng = gpu.RdfEngine(n_per_species, box, cutoff, nbins)
    for rank in range(3):
        eng.accumulate(pos, shard_rank=rank, shard_count=3)
    got = eng.read()
    eng.close()
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
    np.testing.assert_array_equal(got, ref)


def test_two_engines_with_different_lds_needs_alive_at_once(gpu):
    """The kernels' LDS allowance is a property of the kernel, not of the engine: an engine
    with many bins (120 KB of LDS per block) and sorted frames must still launch after an
    engine with few bins and a small species has been created, and the other way round."""
    rng = np.random.default_rng(5)
    big_n, big_box, big_cut, big_bins = [2600, 2300], [30.0] * 3, np.float32(14.9), 14900
    small_n, small_box, small_cut, small_bins = [300], [12.0] * 3, np.float32(5.0), 50
    big_pos = [(rng.random((n, 2, 3)) * 30.0).astype(np.float32) for n in big_n]
    small_pos = [(rng.random((n, 2, 3)) * 12.0).astype(np.float32) for n in small_n]
    big = gpu.RdfEngine(big_n, big_box, big_cut, big_bins)
    small = gpu.RdfEngine(small_n, small_box, small_cut, small_bins)
    big.accumulate(big_pos)
    small.accumulate(small_pos)
    big.accumulate(big_pos)
    got_big, got_small = big.read(), small.read()
    big.close()
    small.close()
    ref_big = c_oracle.rdf_hist(np.concatenate(big_pos, 0), big_n, big_box, big_cut, big_bins)
    ref_small = c_oracle.rdf_hist(np.concatenate(small_pos, 0), small_n, small_box, small_cut,
                                  small_bins)
    np.testing.assert_array_equal(got_big, 2 * ref_big)
    np.testing.assert_array_equal(got_small, ref_small)


@pytest.mark.parametrize("nbins", [4000, 4400, 4990])
def test_noted_pairs_are_resolved_in_batches(gpu, nbins):
    """A wave notes its near-edge pairs in a queue of 320 entries and resolves them 64 at a
    time, in the middle of the j loop as soon as a batch is full.  Atoms on a lattice whose
    distances sit on bin edges fill the queues many times over (up to 256 entries in one
    iteration), in sorted frames (5 000 atoms per species) and with every loop variant in
    play; three bin counts: the second with an LDS histogram close to a block's share of a
    compute unit, the third beyond it (seven blocks per unit)."""
    rng = np.random.default_rng(nbins)
    cutoff = np.float32(20.0)          # bin width 0.005 / 0.004545: lattice distances on edges
    box, n_per_species = [44.0] * 3, [5000, 4100]
    grid = np.arange(0, 88, dtype=np.float32) * np.float32(0.5)
    pos = []
    for n in n_per_species:
        cells = rng.choice(88 ** 3, size=n, replace=False)
        xyz = np.stack([grid[cells % 88], grid[cells // 88 % 88], grid[cells // 7744]], axis=1)
        pos.append(np.repeat(xyz[:, None, :], 2, axis=1).astype(np.float32))
        pos[-1][:, 1] += (rng.random((n, 3)) * 1e-3).astype(np.float32)  # frame 1: just off
    eng = gpu.RdfEngine(n_per_species, box, cutoff, nbins)
    eng.accumulate(pos)
    got = eng.read()
    eng.close()
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
    np.testing.assert_array_equal(got, ref)
    assert ref.sum() > 0


def test_sort_placements_around_their_switch(gpu, monkeypatch):
    """The sort kernel places a species of at most 12 288 atoms through an index list in
    LDS and larger ones wave by wave; a crowded cell sends a small species the second way
    too.  One species either side of the limit, and a species collapsed into one cell,
    with the tile pairs in two shards (both placements must give the order of the input
    within a cell, or the shards' tiles would differ)."""
    monkeypatch.setenv("MDX_RDF_CULL", "1")
    monkeypatch.setenv("MDX_RDF_UNFOLD", "1")
    rng = np.random.default_rng(12288)
    box, cutoff, nbins = [50.0] * 3, np.float32(6.0), 600
    n_per_species = [12290, 12289, 3000]   # strict masks drop one atom each
    pos = [(rng.random((n, 2, 3)) * 50.0).astype(np.float32) for n in n_per_species]
    pos[2][:, :, :] = (np.float32(20.0) + rng.random((3000, 2, 3)) * 1.5).astype(np.float32)
    eng = gpu.RdfEngine(n_per_species, box, cutoff, nbins)
    for rank in range(2):
        eng.accumulate(pos, shard_rank=rank, shard_count=2)
    got = eng.read()
    eng.close()
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, box, cutoff, nbins)
    np.testing.assert_array_equal(got, ref)
    assert ref.sum() > 0


@pytest.mark.parametrize("seed", range(24))
def test_sorted_frames_random_geometries(gpu, monkeypatch, seed):
    """Random systems on the sorted-frame path: one to three species of 2 100 to 5 200
    atoms, cubic and rectangular boxes, cutoffs from a tenth of the shortest edge to just
    below half of it, 50 to 6 000 bins, both mask conventions, uniform, clustered and
    lattice-like positions, every fourth case with a frame whose atoms lie a little outside
    the box.  Sorting and the fold-free loop variants are switched on whatever
    the geometry (the library's own choice would leave out the long boxes), the skipping of
    what the boxes put out of reach for every other case; against the C oracle, bit for bit."""
    monkeypatch.setenv("MDX_RDF_UNFOLD", "1")
    monkeypatch.setenv("MDX_RDF_CULL", "1" if seed % 2 == 0 else "0")
    rng = np.random.default_rng(9000 + seed)
    n_species = int(rng.integers(1, 4))
    n_per_species = [int(rng.integers(2100, 5200)) for _ in range(n_species)]
    edge = float(rng.uniform(30.0, 70.0))
    box = np.array([edge * rng.uniform(1.0, 3.0) if rng.random() < 0.4 else edge
                    for _ in range(3)], np.float32)
    shortest = float(box.min())
    cutoff = np.float32(rng.uniform(0.1 * shortest, 0.5 * shortest - 0.1))
    nbins = int(rng.integers(50, 6001))
    strict = bool(rng.integers(0, 2))
    kind = seed % 3
    pos = []
    for n in n_per_species:
        if kind == 0:
            p = rng.random((n, 2, 3)) * box
        elif kind == 1:
            centres = rng.random((20, 3)) * box
            p = (centres[rng.integers(0, 20, n)][:, None, :]
                 + rng.normal(scale=0.08 * shortest, size=(n, 2, 3))) % box
        else:
            spacing = box / np.float32(32.0)
            p = rng.integers(0, 32, (n, 1, 3)) * spacing + rng.random((n, 2, 3)) * 1e-3
        if seed % 4 == 3:
            # atoms that have left the box since they were last wrapped (second frame only)
            p[:, 1] += rng.normal(scale=0.03 * shortest, size=p[:, 1].shape)
        pos.append(np.ascontiguousarray(p, np.float32))
    eng = gpu.RdfEngine(n_per_species, [float(b) for b in box], cutoff, nbins,
                        strict_reference_masks=strict)
    eng.accumulate(pos)
    got = eng.read()
    eng.close()
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, [float(b) for b in box],
                            cutoff, nbins, strict=strict)
    np.testing.assert_array_equal(got, ref)
    assert ref.sum() > 0


@pytest.mark.parametrize("case", ["a_little_outside", "a_quarter_box_outside", "mixed_frames"])
def test_frames_that_left_the_box_a_little(gpu, monkeypatch, case):
    """Frames whose atoms have moved out of the box since they were last wrapped spread over
    more than one box length; while every coordinate difference stays below the point where
    the literal rule would shift by two box lengths (about 1.5 L) the default kernel takes
    them, sorted like the others.  Atoms up to 0.5 and up to a quarter of a box length outside,
    and a block of frames of all three kinds (inside, a little outside, boxes away: the last
    go to the table kernel); sorted and unsorted, against the C oracle."""
    rng = np.random.default_rng(sum(map(ord, case)))
    n_per_species, frames = [3100, 2700], 4
    box = np.array([38.0, 38.0, 38.0], np.float32)
    cutoff, nbins = np.float32(18.9), 1890
    pos = [(rng.random((n, frames, 3)) * box).astype(np.float32) for n in n_per_species]
    if case == "a_little_outside":
        for p in pos:
            p += rng.normal(scale=0.15, size=p.shape).astype(np.float32)
    elif case == "a_quarter_box_outside":
        for p in pos:
            p[:] = ((rng.random(p.shape) * 1.49 - 0.245) * box).astype(np.float32)
    else:
        for p in pos:
            p[:, 1] += rng.normal(scale=0.2, size=p[:, 1].shape).astype(np.float32)
            p[::7, 2, 0] += np.float32(3.0) * box[0]                    # frame 2: boxes away
            p[:, 3] = ((rng.random(p[:, 3].shape) * 1.45 - 0.2) * box).astype(np.float32)
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, list(box), cutoff, nbins)
    for unfold in ("1", "0"):
        monkeypatch.setenv("MDX_RDF_UNFOLD", unfold)
        monkeypatch.setenv("MDX_RDF_CULL", "0")
        eng = gpu.RdfEngine(n_per_species, list(box), cutoff, nbins)
        eng.accumulate(pos)
        got = eng.read()
        eng.close()
        np.testing.assert_array_equal(got, ref, err_msg=f"unfold={unfold}")
    # short cutoff with skipping on the same frames
    monkeypatch.setenv("MDX_RDF_UNFOLD", "1")
    monkeypatch.setenv("MDX_RDF_CULL", "1")
    short, short_bins = np.float32(5.5), 550
    eng = gpu.RdfEngine(n_per_species, list(box), short, short_bins)
    eng.accumulate(pos)
    got = eng.read()
    eng.close()
    ref = c_oracle.rdf_hist(np.concatenate(pos, 0), n_per_species, list(box), short, short_bins)
    np.testing.assert_array_equal(got, ref)
