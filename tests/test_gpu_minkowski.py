"""The rest of SklearnNN.valid_metrics' Minkowski family (kiez/neighbors/exact/sklearn_nearest_neighbors.py:49 -> scikit-learn's
VALID_METRICS): manhattan = cityblock = l1, chebyshev, minkowski with any p >= 1.  No inner-product form, so the call runs on a
register-tiled VALU kernel and the exact float64 selection (kz_knn.hip: kz_family_dist_kernel / kz_exact_select_kernel) -- against the oracle's restatement of
scikit-learn's DistanceMetric32 / 64 (pinned by tests/golden/f64_manhattan.npz ... f32_cityblock.npz, generated from the real
reference): indices bit-exact, distances to rounding.  `pytest -m gpu`."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# manhattan / chebyshev: the device adds a pair's terms in feature order like scikit-learn -- the same values; minkowski[p]: pow()
# (or a product chain for integer p) against libm's: last-bit differences of the float64 sum, 1e-13 relative is generous;
# float32 inputs: the ranking value is ROUNDED to float32 (DistanceMetric32), a last-bit difference can move it by one float32 ulp
RTOL64, RTOL32 = 1e-13, 2e-7


def _data(rng, n, d, dtype, kind):
    if kind == "uniform":
        return rng.random((n, d)).astype(dtype)
    return rng.standard_normal((n, d)).astype(dtype)


def _rank_tolerant_equal(od, oi, gd, gi, rtol):
    """Indices equal, except inside runs of reference distances that agree to `rtol` (the float32 rounding of the ranking value
    makes true ties; a last-bit difference of the float64 sum before that rounding may swap two rows one float32 ulp apart)."""
    if np.array_equal(oi, gi):
        return True
    for r in np.flatnonzero((oi != gi).any(axis=1)):
        for c in np.flatnonzero(oi[r] != gi[r]):
            if not np.isclose(od[r, c], gd[r, c], rtol=rtol, atol=0):
                return False
    return True


@pytest.mark.parametrize("metric,p", [("manhattan", 2), ("chebyshev", 2), ("minkowski", 3), ("minkowski", 1.5), ("l1", 2),
                                      ("minkowski", float("inf")), ("minkowski", 1)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_knn_against_the_oracle(metric, p, dtype):
    from kiez_amd import _native as N
    from kiez_amd.neighbors import canonical_metric
    from oracle import kiez_oracle as O
    ctx = N.Context.get()
    rng = np.random.default_rng(int(p * 10) if np.isfinite(p) else 99)
    mc = canonical_metric(metric, p)
    assert mc == O.canonical_metric(metric, p)
    for n_q, n_i, d, k, kind in ((257, 1301, 33, 10, "gauss"), (64, 5000, 300, 50, "uniform"), (100, 700, 5, 7, "gauss"),
                                 (31, 300, 513, 3, "uniform")):
        q, y = _data(rng, n_q, d, dtype, kind), _data(rng, n_i, d, dtype, kind)
        dd, ii, st = N.knn(ctx, N.DeviceMatrix(ctx, q, mc), N.DeviceMatrix(ctx, y, mc), k)
        assert st["n_fallback_rows"] == n_q, st      # every row on the exact kernels, no MFMA pass
        od, oi = O.knn_exact(q, y, k, mc)
        rtol = RTOL32 if dtype == np.float32 else RTOL64
        assert _rank_tolerant_equal(od, oi, dd.numpy(), ii.numpy(), rtol), (metric, p, dtype, d)
        assert (ii.numpy() == oi).mean() > 0.999
        np.testing.assert_allclose(dd.numpy(), od, rtol=rtol, atol=0)
        if mc in ("manhattan", "chebyshev"):     # same terms, same order of additions as scikit-learn: bit for bit
            np.testing.assert_array_equal(ii.numpy(), oi)
            np.testing.assert_array_equal(dd.numpy(), od)


@pytest.mark.parametrize("metric,p", [("manhattan", 2), ("minkowski", 3), ("chebyshev", 2)])
def test_self_query_drops_the_row_itself(metric, p):
    """kneighbors() of a single-source fit (sklearn/neighbors/_base.py:828-834, 937-965)."""
    from kiez_amd import _native as N
    from kiez_amd.neighbors import canonical_metric
    from oracle import kiez_oracle as O
    ctx = N.Context.get()
    rng = np.random.default_rng(3)
    y = rng.standard_normal((900, 40)).astype(np.float32)
    y[17] = y[400]       # exact duplicates: the self entry is not necessarily the first one
    mc = canonical_metric(metric, p)
    m = N.DeviceMatrix(ctx, y, mc)
    dd, ii, _ = N.knn(ctx, m, m, 6, exclude_self=True)
    od, oi = O.knn_exact(y, y, 6, mc, exclude_self=True)
    np.testing.assert_array_equal(ii.numpy(), oi)
    np.testing.assert_allclose(dd.numpy(), od, rtol=RTOL32, atol=0)
    assert not (ii.numpy() == np.arange(900)[:, None]).any()


@pytest.mark.parametrize("hub,kw", [(None, {}), ("CSLS", {}), ("LocalScaling", {"method": "nicdm"}), ("MutualProximity", {"method": "normal"}),
                                    ("MutualProximity", {"method": "empiric"})])
@pytest.mark.parametrize("metric,p,dtype", [("manhattan", 2, np.float64), ("minkowski", 3, np.float32), ("chebyshev", 2, np.float32)])
def test_kiez_pipeline_against_the_oracle(hub, kw, metric, p, dtype):
    from kiez_amd import Kiez
    from oracle import kiez_oracle as O
    from tests.golden_util import knife_edge_rows, knife_edge_topk_ok
    rng = np.random.default_rng(11)
    s, t = rng.random((700, 48)).astype(dtype), rng.random((900, 48)).astype(dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=10, algorithm="SklearnNN", algorithm_kwargs={"metric": metric, "p": p}, hubness=hub, hubness_kwargs=dict(kw))
        d, i = kz.fit(s, t).kneighbors(5)
    od, oi, inter = O.kiez_pipeline(s, t, 10, 5, metric, p, hub, kw, return_intermediates=True)
    keep = np.ones(len(i), dtype=bool)
    if kw.get("method") == "empiric":
        keep &= ~knife_edge_rows(inter["ind_s2t"])
        for r in np.flatnonzero(~keep):
            assert knife_edge_topk_ok(od[r], oi[r], d[r], i[r], r, 10, inter["ind_t2s"]), r
    np.testing.assert_array_equal(i[keep], oi[keep])
    np.testing.assert_allclose(d[keep], od[keep], rtol=1e-5, atol=1e-9)      # north-star tolerance for rescaled distances


def test_dis_sim_local_rejects_the_family_like_the_reference():
    from kiez_amd import Kiez
    for akw in ({"metric": "manhattan"}, {"metric": "minkowski", "p": 3}, {"metric": "chebyshev"}):
        with pytest.raises(ValueError, match="only supports"):     # kiez/hubness_reduction/dis_sim.py:47-61
            Kiez(algorithm="SklearnNN", algorithm_kwargs=akw, hubness="DisSimLocal")


def test_exponent_must_agree_and_be_at_least_one():
    from kiez_amd import _native as N
    ctx = N.Context.get()
    y = np.random.default_rng(0).random((300, 20))
    a, b = N.DeviceMatrix(ctx, y, "minkowski[3.0]"), N.DeviceMatrix(ctx, y, "minkowski[4.0]")
    with pytest.raises(Exception, match="different metrics"):
        N.knn(ctx, a, b, 3)
    with pytest.raises(Exception, match="p must be >= 1"):
        N.DeviceMatrix(ctx, y, "minkowski[0.5]")
    e = N.DeviceMatrix(ctx, y, "euclidean")
    assert ctx.lib.kz_matrix_set_minkowski_p(e.handle, 3.0) != 0     # only a KZ_MINKOWSKI matrix has an exponent


def test_pair_values_are_the_values_the_search_ranked_by():
    """kz_pair_values with the exponent (the ordering values that travel with a shard's reverse lists across GPUs): bit for bit
    the values behind the neighbours kz_knn returned -- they ascend along every row and map to the returned distances."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    ctx = N.Context.get()
    rng = np.random.default_rng(2)
    q, y = rng.standard_normal((200, 30)).astype(np.float32), rng.standard_normal((500, 30)).astype(np.float32)
    for mc in ("minkowski[3.0]", "manhattan", "chebyshev"):
        qm, ym = N.DeviceMatrix(ctx, q, mc), N.DeviceMatrix(ctx, y, mc)
        d, i, _ = N.knn(ctx, qm, ym, 8)
        val = ctx.empty((200, 8), np.float64)
        N._check(ctx.lib.kz_pair_values(ctx.handle, qm.handle, 0, 200, ym.handle, i.ptr, 8, val.ptr), "kz_pair_values")
        v = val.numpy()
        assert (np.diff(v, axis=1) >= 0).all()
        out = (v ** (1.0 / 3.0)).astype(np.float32).astype(np.float64) if mc.startswith("minkowski[") else v
        np.testing.assert_allclose(d.numpy(), out, rtol=1e-7 if mc.startswith("minkowski[") else 0, atol=0)
        want = np.take_along_axis(O.minkowski_family_rdist(q, y, mc), i.numpy(), axis=1)
        np.testing.assert_allclose(v, want, rtol=RTOL32, atol=0)
