"""Multi-rank paths through the real kernels on one GPU.

Two ranks share GPU 0 (the host transport carries the sums then; RCCL refuses two ranks
on one device), each running the calculators of the product API on its shard: atom-pair
tiles for the RDF, frames for the ionic-current reduction, window origins for the
correlations.  The results must equal the one-rank run: bit for bit for the integer
histograms (hence for g(r)), to 1e-12 of the largest value for the float64 correlations
(the partial sums are added in a different order).  A one-rank RCCL communicator checks
the device collective itself (librccl loading, communicator creation, in-place uint64 and
float64 all-reduce, ``mdx_rdf_reduce``).
"""
import numpy as np
import pytest

import _rank_workers as rw

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.max(np.abs(a - b)) <= tol * np.max(np.abs(b))


def test_two_ranks_on_one_gpu_match_one_rank(gpu, tmp_path):
    (tmp_path / "one").mkdir()
    (tmp_path / "two").mkdir()
    one = rw.run_ranks(rw.gpu_calculators, 1, str(tmp_path / "one"))[0][1]
    two = rw.run_ranks(rw.gpu_calculators, 2, str(tmp_path / "two"))
    assert one["transport"] == "none"
    for rank, res in two:
        assert res["transport"] == "host"
        assert set(res["rdf"]) == set(one["rdf"]) == {"Na_Na", "Na_Cl", "Cl_Cl"}
        for name in ("rdf", "rdf_tiles", "rdf_frames"):  # auto, tile shards, frame shards
            for key in one[name]:
                np.testing.assert_array_equal(res[name][key], one[name][key])
        _close(res["gk_acf"], one["gk_acf"])
        _close(res["gk_sigma"], one["gk_sigma"], 1e-10)
        _close(res["gk_unc"], one["gk_unc"], 1e-9)
        for key in one["vacf"]:
            _close(res["vacf"][key], one["vacf"][key])
        assert not [f for f in res["files"] if f.endswith(".tmp")]
    # both ranks hold the same results
    np.testing.assert_array_equal(two[0][1]["gk_acf"], two[1][1]["gk_acf"])


def test_one_rank_rccl_communicator(gpu):
    res = rw.run_ranks(rw.rccl_single_rank, 1, transport="rccl")[0][1]
    assert res["transport"] == "rccl"
    assert res["u64_ok"] and res["f64_ok"] and res["max_ok"] and res["rdf_ok"]
