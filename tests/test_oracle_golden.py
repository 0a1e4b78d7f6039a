"""The oracle (oracle/kiez_oracle.py) against golden vectors produced by the real reference
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import kiez_oracle as O
from tests.golden_util import HUB, case_params, ktag, load_case

RTOL, ATOL = 1e-9, 5e-7  # ATOL covers the self distances of a single-source reverse pass: exact 0 vs sqrt(1e-14) from the expanded form


@pytest.mark.parametrize("case,tag,k", case_params())
def test_pipeline_matches_reference(case, tag, k):
    g = load_case(case)
    hname, kw = HUB[tag]
    d, i, inter = O.kiez_pipeline(g["source"], g["_target"], g["_K"], k, g["_metric"], g["_p"], hname, kw,
                                  return_intermediates=True)
    ref_d, ref_i = g[f"{tag}__k{ktag(k)}__dist"], g[f"{tag}__k{ktag(k)}__ind"]
    assert i.dtype == np.int64 and ref_i.shape == i.shape
    np.testing.assert_array_equal(i, ref_i)
    rtol, atol = RTOL, ATOL
    if tag == "mp_normal":
        rtol = 1e-7
    if tag == "dsl":
        # float32 inputs: the reference evaluates DSL partly in float32 (sklearn euclidean_distances /
        # einsum on float32), the oracle in float64 -> the north-star tolerance 1e-5 applies
        rtol, atol = (1e-5, ATOL) if g["source"].dtype == np.float32 else (1e-7, ATOL)
    np.testing.assert_allclose(d, ref_d, rtol=rtol, atol=atol)
    if tag != "none":
        np.testing.assert_array_equal(inter["ind_t2s"], g[f"{tag}__ind_t2s"])
        np.testing.assert_array_equal(inter["ind_s2t"], g[f"{tag}__ind_s2t"])
        np.testing.assert_allclose(inter["dist_t2s"], g[f"{tag}__dist_t2s"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(inter["dist_s2t"], g[f"{tag}__dist_s2t"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(inter["transformed"], g[f"{tag}__transformed"], rtol=rtol, atol=atol)


def test_dsl_cosine_raises_like_reference():
    g = load_case("cosine_k50")
    assert "dsl__raises" in g and "DisSimLocal only supports squared Euclidean" in str(g["dsl__raises"])
    with pytest.raises(ValueError):
        O.kiez_pipeline(g["source"], g["_target"], 10, 5, "cosine", 2, "DisSimLocal", {})


@pytest.mark.parametrize("k", [1, 2, 3, 5, 10])
def test_sort_topk_matches_reference_sort(k):
    z = np.load(load_case.__globals__["GOLDEN"] / "sort.npz")
    d, i = O.sort_topk(z["dist0"], z["ind0"], k)
    np.testing.assert_array_equal(i, z[f"sorted0_k{k}_ind"])
    np.testing.assert_array_equal(d, z[f"sorted0_k{k}_dist"])
    for pre, dd in (("tie", z["tie_d"]), ("tie32", z["tie_d"].astype(np.float32))):
        d, i = O.sort_topk(dd, z["tie_i"], k)
        np.testing.assert_array_equal(d, z[f"{pre}_k{k}_dist"])
        np.testing.assert_array_equal(i, z[f"{pre}_k{k}_ind"])


def test_sort_topk_is_numpy_argpartition_for_k_ge_2():
    """for k>=2 numpy's argpartition(kth=arange(k)) is a selection sort with swaps."""
    rng = np.random.RandomState(123)
    for K, k in ((12, 2), (12, 7), (50, 50), (33, 5)):
        d = rng.randint(0, 5, size=(500, K)).astype(np.float64)
        ind = rng.randint(0, 10**6, size=(500, K)).astype(np.int64)
        mask = np.argpartition(d, kth=np.arange(k))[:, :k]
        od, oi = O.sort_topk(d, ind, k)
        np.testing.assert_array_equal(np.take_along_axis(ind, mask, axis=1), oi)
        np.testing.assert_array_equal(np.take_along_axis(d, mask, axis=1), od)


def test_knn_exact_against_sklearn_brute():
    """The restated kNN against the third-party code the reference actually calls."""
    from sklearn.neighbors import NearestNeighbors
    rng = np.random.RandomState(1)
    x = rng.rand(500, 40).astype(np.float32)
    y = rng.rand(700, 40).astype(np.float32)
    for metric in ("euclidean", "sqeuclidean"):
        nn = NearestNeighbors(n_neighbors=10, algorithm="brute", metric=metric).fit(y)
        d, i = nn.kneighbors(x)
        od, oi = O.knn_exact(x, y, 10, metric)
        np.testing.assert_array_equal(i, oi)
        np.testing.assert_allclose(d, od, rtol=1e-9, atol=1e-9)
    nn = NearestNeighbors(n_neighbors=10, algorithm="brute", metric="cosine").fit(y.astype(np.float64))
    d, i = nn.kneighbors(x.astype(np.float64))
    od, oi = O.knn_exact(x, y, 10, "cosine")
    np.testing.assert_array_equal(i, oi)
    np.testing.assert_allclose(d, od, rtol=1e-9, atol=1e-12)
    nn = NearestNeighbors(n_neighbors=10, algorithm="brute", metric="euclidean").fit(y)
    d, i = nn.kneighbors()
    od, oi = O.knn_exact(y, y, 10, "euclidean", exclude_self=True)
    np.testing.assert_array_equal(i, oi)
    np.testing.assert_allclose(d, od, rtol=1e-9, atol=1e-7)


def test_oracle_orders_near_ties_as_the_reference_from_two_ulps_on():
    """tests/golden/near_ties.npz (tools/gen_near_ties.py): pairs of index rows 1/64 .. 16 ulps apart.  From two ulps (of
    |q|^2 + |y|^2) on, the oracle's float64 expansion must order every pair as the reference did and as exact arithmetic does;
    below, the order is a property of the summation order (largest gap the oracle gets wrong: 1.04 ulps, the reference: 0.97)."""
    import numpy as np
    from pathlib import Path
    from oracle import kiez_oracle as O
    fx = np.load(Path(__file__).resolve().parent / "golden" / "near_ties.npz")
    _, ind = O.knn_exact(fx["query"], fx["index"], 2, "sqeuclidean")
    assert (np.sort(ind, axis=1) == np.sort(fx["ref_ind"], axis=1)).all()
    sel = fx["gap_ulps"] >= 2.0
    assert sel.sum() > 250
    np.testing.assert_array_equal(ind[sel, 0], fx["ref_ind"][sel, 0])
    np.testing.assert_array_equal(ind[sel, 0], fx["exact_nearer"][sel])
