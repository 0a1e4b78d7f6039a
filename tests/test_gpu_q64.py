"""The 64-queries-per-wave build of the fp16 kernel (kz_knn_h64.h, kz_knn_epi4.h), forced on every shape it is built for
(`h_q64 = 1`; by default it only takes the shared sweep of large launches): ordinary searches against the oracle and against the
32-query kernel, the shared sweep against two searches, units that reach past the last query tile, single-source mode, every
slice count from 4 to 13, exact ties, the golden vectors through the API.  Reference path: kiez/neighbors/exact/
sklearn_nearest_neighbors.py:96-101 (kneighbors), kiez/hubness_reduction/base.py:33-50 (both directions of a fit)."""
import warnings

import numpy as np
import pytest

from tests.test_gpu_dual import _assert_same, _both_ways, _data, _oracle_sample

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    c = N.Context.get()
    c.set_option("h_q64", 1)
    yield c
    for name, value in (("h_q64", 2), ("dual_force", 0), ("force_splits", 0), ("chunk_rows", 0)):
        c.set_option(name, value)


@pytest.mark.parametrize("d", [50, 64, 72, 96, 100, 128, 130, 160, 176, 200, 208])   # 4 .. 13 slices
def test_every_slice_count_against_the_oracle_and_the_32_query_kernel(ctx, d):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(d)
    q, y = rng.rand(3000 + d, d).astype(np.float32), rng.rand(20000 + 3 * d, d).astype(np.float32)   # ragged last tiles, odd tile counts
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    dist, ind, st = N.knn(ctx, qm, ym, 10)
    assert st["first_pass"] == 2 and st["n_fallback_rows"] == st["n_spec_rows"] and st["max_err_ratio"] < 1.0
    od, oi = O.knn_exact(q, y, 10, "euclidean")
    np.testing.assert_array_equal(ind.numpy(), oi)
    np.testing.assert_array_equal(dist.numpy(), od)
    ctx.set_option("h_q64", 0)
    d0, i0, _ = N.knn(ctx, qm, ym, 10)
    np.testing.assert_array_equal(ind.numpy(), i0.numpy())
    np.testing.assert_array_equal(dist.numpy(), d0.numpy())


@pytest.mark.parametrize("n_q", [1, 127, 128, 129, 256, 257, 384, 385, 1000])   # units of two query tiles: the second one missing / ragged
def test_units_that_reach_past_the_last_query_tile(ctx, n_q):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(n_q)
    q, y = rng.rand(n_q, 128).astype(np.float32), rng.rand(9000, 128).astype(np.float32)
    dist, ind, st = N.knn(ctx, N.DeviceMatrix(ctx, q, "cosine"), N.DeviceMatrix(ctx, y, "cosine"), 7)
    od, oi = O.knn_exact(q.astype(np.float64), y.astype(np.float64), 7, "cosine")
    np.testing.assert_array_equal(ind.numpy(), oi)
    np.testing.assert_allclose(dist.numpy(), od, rtol=1e-5, atol=1e-7)


def test_many_index_ranges_per_query_tile_and_chunked_launches(ctx):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(3)
    q, y = rng.rand(5000, 100).astype(np.float32), rng.rand(40000, 100).astype(np.float32)
    qm, ym = N.DeviceMatrix(ctx, q, "sqeuclidean"), N.DeviceMatrix(ctx, y, "sqeuclidean")
    od, oi = O.knn_exact(q, y, 10, "sqeuclidean")
    for splits, chunk in ((7, 0), (1, 1024), (0, 640)):
        ctx.set_option("force_splits", splits)
        ctx.set_option("chunk_rows", chunk)
        dist, ind, _ = N.knn(ctx, qm, ym, 10)
        np.testing.assert_array_equal(ind.numpy(), oi, err_msg=f"splits {splits} chunk {chunk}")
        np.testing.assert_allclose(dist.numpy(), od, rtol=1e-12, atol=1e-12)   # (unrounded float64 values of two summation orders)


def test_single_source_mode_strips_the_row_itself(ctx):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(5)
    x = rng.rand(7001, 64).astype(np.float32)
    xm = N.DeviceMatrix(ctx, x, "euclidean")
    dist, ind, _ = N.knn(ctx, xm, xm, 10, exclude_self=True)
    od, oi = O.knn_exact(x, x, 10, "euclidean", exclude_self=True)
    np.testing.assert_array_equal(ind.numpy(), oi)
    np.testing.assert_array_equal(dist.numpy(), od)


@pytest.mark.parametrize("kind,na,nb,d,k,metric,dtype", [
    ("uniform", 20000, 6000, 64, 10, "euclidean", np.float32),
    ("uniform", 9000, 30011, 72, 5, "sqeuclidean", np.float64),      # odd slice count, a smaller than b, float64
    ("uniform", 16384, 4096, 200, 10, "euclidean", np.float32),      # 13 slices: the headline kernel's build
    ("normal", 12000, 8000, 200, 50, "cosine", np.float32),          # short-list route: ten lists of 16 per query
    ("clustered", 20000, 12000, 80, 10, "euclidean", np.float32),    # hubs: bursts of column events
    ("duplicates", 10000, 7000, 96, 10, "sqeuclidean", np.float32),  # exact ties in both directions
    ("uniform", 5001, 1029, 128, 3, "cosine", np.float64),           # ragged last tiles, odd unit count
])
def test_shared_sweep_identical_to_two_searches(ctx, kind, na, nb, d, k, metric, dtype):
    ctx.set_option("dual_force", 1)
    a, b = _data(kind, na, d, 1, dtype), _data(kind, nb, d, 2, dtype)
    sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, k, metric)
    _assert_same(sep, dual)
    if kind != "duplicates":
        _oracle_sample(a, b, k, metric, dual)
        assert s_ab["dual"] == 1
    assert s_ab["max_err_ratio"] < 1.0 and s_ba["max_err_ratio"] < 1.0


@pytest.mark.parametrize("hub,kw", [("CSLS", {}), ("LocalScaling", {"method": "nicdm"}), ("MutualProximity", {"method": "normal"})])
def test_api_through_the_shared_sweep(ctx, hub, kw):
    from kiez_amd import Kiez
    from oracle import kiez_oracle as O
    ctx.set_option("dual_force", 1)
    rng = np.random.RandomState(11)
    s, t = rng.rand(6000, 64).astype(np.float32), rng.rand(5000, 64).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=10, algorithm="SklearnNN", algorithm_kwargs={"metric": "euclidean"}, hubness=hub, hubness_kwargs=kw)
        dist, ind = kz.fit(s, t).kneighbors(5)
        assert kz.algorithm.last_stats["dual"] == 1
    od, oi = O.kiez_pipeline(s, t, 10, 5, "euclidean", 2, hub, kw)
    np.testing.assert_array_equal(ind, oi)
    np.testing.assert_allclose(dist, od, rtol=1e-5, atol=1e-6)
