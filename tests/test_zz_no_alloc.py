"""The entry point that allocates nothing (``mdx_weighted_atom_sum_static``) against the
allocating one (``mdx_weighted_atom_sum``), through the C-ABI, on two calls in a row, the
second with a smaller input; then sizes that make it cap its chunk count, the accumulate
mode, float64 input and the size limit."""
import numpy as np
import pytest

from lammps_analysis_amd import _lib

pytestmark = pytest.mark.gpu


def _static_sum(engine, dev_x, n_atoms, n_frames, k, weight, out_dev, accumulate=False):
    code = _lib.MDX_F32 if dev_x.dtype == np.float32 else _lib.MDX_F64
    _lib.check(_lib.load().mdx_weighted_atom_sum_static(
        dev_x.ptr, code, n_atoms, n_frames, k, float(weight), out_dev.ptr,
        1 if accumulate else 0))


def _compare(engine, rng, n_atoms, n_frames, k, weight, dtype=np.float32):
    x = rng.normal(size=(n_atoms, n_frames, k)).astype(dtype)
    dev_x = engine.DeviceArray.from_numpy(x)
    ref_dev = engine.DeviceArray((n_frames, k), np.float64)
    engine.weighted_atom_sum(dev_x, scalar_weight=weight, out=ref_dev)  # allocating path
    ref = ref_dev.to_numpy()
    out_dev = engine.DeviceArray((n_frames, k), np.float64)
    _static_sum(engine, dev_x, n_atoms, n_frames, k, weight, out_dev)
    got = out_dev.to_numpy()
    # float64 sums of the same products; only the chunking of the atom axis may differ
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * max(1.0, np.max(np.abs(ref))))
    np.testing.assert_allclose(ref, weight * x.astype(np.float64).sum(axis=0), rtol=0,
                               atol=1e-9 * max(1.0, np.max(np.abs(ref))))
    return dev_x, out_dev, got


def test_static_entry_point_matches_allocating_one(gpu):
    engine = gpu
    rng = np.random.default_rng(77)
    # two calls in a row sharing the static array, the second one smaller
    _, _, first = _compare(engine, rng, 700, 4096, 3, -1.5)
    _, _, second = _compare(engine, rng, 90, 333, 3, 2.0)
    assert first.shape == (4096, 3) and second.shape == (333, 3)
    # unaligned rows (scalar loads), float64 input, one atom, one frame
    _compare(engine, rng, 33, 7, 1, 1.0)
    _compare(engine, rng, 257, 1000, 3, 0.25, dtype=np.float64)
    _compare(engine, rng, 1, 5, 3, 3.0)
    _compare(engine, rng, 40, 1, 2, 1.0)


def test_static_entry_point_accumulates_and_caps_its_chunks(gpu):
    engine = gpu
    rng = np.random.default_rng(78)
    # 1.2 M values per row: only three chunks of partial sums fit the static array
    n_atoms, n_frames, k = 64, 400000, 3
    x = rng.normal(size=(n_atoms, n_frames, k)).astype(np.float32)
    dev_x = engine.DeviceArray.from_numpy(x)
    base = rng.normal(size=(n_frames, k))
    out_dev = engine.DeviceArray.from_numpy(base)
    _static_sum(engine, dev_x, n_atoms, n_frames, k, 1.0, out_dev, accumulate=True)
    ref_dev = engine.DeviceArray.from_numpy(base)
    engine.weighted_atom_sum(dev_x, scalar_weight=1.0, out=ref_dev, accumulate=True)
    ref = ref_dev.to_numpy()
    np.testing.assert_allclose(out_dev.to_numpy(), ref, rtol=0,
                               atol=1e-12 * np.max(np.abs(ref)))


def test_static_entry_point_rejects_rows_beyond_its_array(gpu):
    engine = gpu
    dev_x = engine.DeviceArray((1, 8), np.float32)  # never read: the size check comes first
    out_dev = engine.DeviceArray((1, 8), np.float64)
    with pytest.raises(ValueError, match="exceeds"):
        _lib.check(_lib.load().mdx_weighted_atom_sum_static(
            dev_x.ptr, _lib.MDX_F32, 2, (4 << 20) // 3 + 1, 3, 1.0, out_dev.ptr, 0))
    # empty input: zero result, nothing launched
    out = engine.DeviceArray.from_numpy(np.ones((5, 3)))
    _lib.check(_lib.load().mdx_weighted_atom_sum_static(dev_x.ptr, _lib.MDX_F32, 0, 5, 3, 1.0,
                                                        out.ptr, 0))
    assert not out.to_numpy().any()


def _counters():
    import ctypes
    a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _lib.check(_lib.load().mdx_allocation_counters(ctypes.byref(a), ctypes.byref(b),
                                                   ctypes.byref(c)))
    return a.value, b.value, c.value


def test_static_entry_point_allocates_nothing(gpu):
    """After the first call on the device (which looks up the array's address), further
    calls of any admissible size make no request to the driver's allocators or to the
    library's block cache, and the device's free memory does not move; the allocating
    entry point, called between them, does show up in the same counters."""
    engine = gpu
    rng = np.random.default_rng(79)
    shapes = [(300, 2048, 3), (17, 100, 1), (64, 400000, 3), (5, 64, 2)]
    inputs = []
    for n_atoms, n_frames, k in shapes:
        x = engine.DeviceArray.from_numpy(
            rng.normal(size=(n_atoms, n_frames, k)).astype(np.float32))
        inputs.append((x, engine.DeviceArray((n_frames, k), np.float64)))
    _static_sum(engine, inputs[0][0], *shapes[0], 1.0, inputs[0][1])   # first call: set-up
    engine.synchronize()
    before = _counters()
    for _ in range(3):
        for (x, out), (n_atoms, n_frames, k) in zip(inputs, shapes):
            _static_sum(engine, x, n_atoms, n_frames, k, 0.5, out)
            _static_sum(engine, x, n_atoms, n_frames, k, 0.5, out, accumulate=True)
    engine.synchronize()
    after = _counters()
    assert after[0] == before[0], "a static call reached hipMalloc / hipHostMalloc"
    assert after[1] == before[1], "a static call asked the library's block cache for memory"
    assert after[2] == before[2], "the device's free memory moved"
    x, out = inputs[0]
    np.testing.assert_allclose(out.to_numpy(),
                               x.to_numpy().astype(np.float64).sum(axis=0), atol=1e-9)
    # the counters do see the allocating path
    engine.weighted_atom_sum(x, scalar_weight=1.0, out=out)
    assert _counters()[1] > after[1]
