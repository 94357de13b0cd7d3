"""Generates the golden fixtures in this directory.

The reference itself cannot be imported in the build container (TensorFlow,
TensorFlow-Probability, JAX and h5py are not installed), so the expected outputs come
from the CPU oracle (oracle/rdf.py, oracle/correlation.py), which restates the
reference op by op.  Every fixture stores inputs, arguments and expected outputs, so
the GPU box needs nothing but these files.  Run from the repository root:

    python tests/golden/make_golden.py
"""
import json
import pathlib
import sys

import numpy as np
from scipy.integrate import cumulative_trapezoid

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import correlation as orc  # noqa: E402
from oracle import rdf as orc_rdf  # noqa: E402

HERE = pathlib.Path(__file__).resolve().parent


def rdf_2sp_small():
    rng = np.random.default_rng(0)
    box = np.array([12.0, 12.0, 12.0])
    cutoff = np.float32(box[0] / 2 - 0.1)
    nbins = int(cutoff / 0.01)
    pos = [(rng.random((64, 4, 3)) * box).astype(np.float32) for _ in range(2)]
    hist = orc_rdf.rdf_histograms(pos, ["Na", "Cl"], box, cutoff, nbins, rdf_minibatch=40)
    np.savez_compressed(HERE / "rdf_2sp_small.npz", pos_na=pos[0], pos_cl=pos[1], box=box,
                        cutoff=cutoff, nbins=nbins,
                        hist=np.stack([hist[k] for k in orc_rdf.key_names(["Na", "Cl"])]))


def rdf_edge():
    """Pairs on bin edges, at the cutoff, at half the box (rint ties), many images away."""
    box = np.array([10.0, 10.0, 10.0], dtype=np.float32)
    cutoff = np.float32(4.9)
    nbins = 490
    step = float(cutoff) / nbins
    others = []
    for k in range(0, nbins + 3, 7):
        edge = np.float32(k * step)
        for value in (np.nextafter(edge, np.float32(-1)), edge, np.nextafter(edge, np.float32(99))):
            others.append([value, 0.0, 0.0])
    for d in (5.0, np.nextafter(np.float32(5.0), np.float32(0)),
              np.nextafter(np.float32(5.0), np.float32(9)), -5.0, 15.0, 25.0, 4.9, 1.0e4):
        others.append([0.0, d, 0.0])
    others = np.asarray(others, dtype=np.float32)
    a = np.concatenate([np.full((1, 3), 3.3, np.float32), np.zeros((1, 3), np.float32)])[:, None]
    b = np.concatenate([np.full((1, 3), 7.7, np.float32), others])[:, None]
    pos = [np.ascontiguousarray(a), np.ascontiguousarray(b)]
    hist = orc_rdf.rdf_histograms(pos, ["A", "B"], box, cutoff, nbins)
    np.savez_compressed(HERE / "rdf_edge.npz", pos_a=pos[0], pos_b=pos[1], box=box,
                        cutoff=cutoff, nbins=nbins,
                        hist=np.stack([hist[k] for k in orc_rdf.key_names(["A", "B"])]))


def _window_table(x, d, ct, n_win):
    rows = []
    for w in range(n_win):
        win = x[:, w * ct:w * ct + d]
        rows.append(np.sum(orc.auto_correlation(win, axis=1), axis=(0, 2)))
    return np.array(rows)


def acf_sys():
    rng = np.random.default_rng(1)
    T, d = 2048, 64
    x = np.cumsum(rng.normal(size=(1, T, 3)), axis=1) * 0.05 + rng.normal(size=(1, T, 3))
    x = x.astype(np.float32)
    time = np.arange(d) * 0.002
    out = {"x": x, "data_range": d, "time": time}
    for ct in (1, 7):
        n_win = int(np.clip((T - d) / ct, 1, None))
        table = _window_table(x.astype(np.float64), d, ct, n_win)
        sig = np.array([cumulative_trapezoid(r, x=time) for r in table])
        out[f"acf_sum_ct{ct}"] = table.sum(axis=0)
        out[f"sigma_mean_ct{ct}"] = sig.mean(axis=0)
        out[f"sigma_std_ct{ct}"] = sig.std(axis=0)
        out[f"n_win_ct{ct}"] = n_win
    np.savez_compressed(HERE / "acf_sys.npz", **out)


def vacf_atoms():
    rng = np.random.default_rng(2)
    n, T, d = 32, 1024, 128
    v = rng.normal(size=(n, T, 3)).astype(np.float32)
    n_win = T - d
    table = _window_table(v.astype(np.float64), d, 1, n_win)
    np.savez_compressed(HERE / "vacf_atoms.npz", v=v, data_range=d, n_win=n_win,
                        acf_sum=table.sum(axis=0))


def msd_atoms():
    rng = np.random.default_rng(3)
    n, T, d = 24, 600, 100
    pos = np.cumsum(rng.normal(size=(n, T, 3)), axis=1).astype(np.float32)
    out = {"pos": pos, "data_range": d}
    for ct in (1, 3):
        msd_sum, count, tau = orc.einstein_msd(pos, d, ct)
        out[f"msd_sum_ct{ct}"] = msd_sum
        out[f"count_ct{ct}"] = count
    np.savez_compressed(HERE / "msd_atoms.npz", **out)


def batching():
    cases = []
    for n_bytes, n_configs, memory, fraction, factor in (
            ([500], 10, 50000, 0.5, 10), ([500], 13, 50, 1.0, 10),
            ([864 * 1000 * 12], 1000, 30e9, 0.5, 150), ([100000 * 12], 100000, 30e9, 0.5, 5),
            ([100 * 2645 * 12], 2645, 1.0737e6, 1.0, 150)):
        cases.append({"n_bytes": n_bytes, "n_configs": n_configs, "memory": memory,
                      "fraction": fraction, "scale_factor": factor,
                      "expected": list(orc.get_batch_size(
                          n_bytes, n_configs, memory, fraction, orc.linear_scale_function,
                          {"scale_factor": factor}))})
    windows = [{"data_size": s, "data_range": d, "correlation_time": c,
                "n_windows": len(orc.window_starts(s, d, c))}
               for s, d, c in ((1000, 100, 1), (100000, 500, 1), (50, 10, 5), (5, 10, 5),
                               (2048, 64, 7))]
    with open(HERE / "batching.json", "w") as fh:
        json.dump({"batch_size": cases, "windows": windows}, fh, indent=1)


if __name__ == "__main__":
    rdf_2sp_small()
    rdf_edge()
    acf_sys()
    vacf_atoms()
    msd_atoms()
    batching()
    print("golden fixtures written to", HERE)
