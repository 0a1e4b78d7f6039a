"""The multi-GPU exchange kernels (kz_pair_values, kz_merge_topk) and the kernels for more than 128 candidates per query
(wide kz_select_topk / kz_mp_empiric; the reference has no cap on n_candidates, kiez/hubness_reduction/base.py:20-27).
Needs an MI355X: `pytest -m gpu`."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    return N.Context.get()


@pytest.mark.parametrize("metric,dtype,d", [("euclidean", np.float32, 200), ("sqeuclidean", np.float64, 37), ("cosine", np.float32, 300),
                                            ("cosine", np.float64, 64), ("euclidean", np.float64, 515)])
def test_pair_values_are_the_values_the_search_ranked_by(ctx, metric, dtype, d):
    """kz_pair_values must reproduce, bit for bit, the value behind every neighbour kz_knn returned: pushed through
    sklearn's output rule it IS the returned distance, and it ascends along every row (ties: ascending index)."""
    from kiez_amd import _native as N
    rng = np.random.default_rng(3)
    q = rng.random((3000, d)).astype(dtype)
    y = rng.random((5000, d)).astype(dtype)
    y[100:110] = y[90:100]                       # exact duplicates: tied values
    qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
    k = 12
    dist, ind, _ = N.knn(ctx, qm, ym, k)
    val = ctx.empty((1000, k), np.float64)
    sub = ind.view_rows(500, 1000)
    N._check(ctx.lib.kz_pair_values(ctx.handle, qm.handle, 500, 1000, ym.handle, sub.ptr, k, val.ptr), "kz_pair_values")
    v, dd, ii = val.numpy(), dist.numpy()[500:1500], ind.numpy()[500:1500]
    if metric == "euclidean" and dtype == np.float32:
        out = np.sqrt(v.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)
    elif metric == "euclidean":
        out = np.sqrt(v)
    else:
        out = v
    np.testing.assert_array_equal(out, dd)
    asc = (v[:, 1:] > v[:, :-1]) | ((v[:, 1:] == v[:, :-1]) & (ii[:, 1:] > ii[:, :-1]))
    assert asc.all()
    # symmetric: the same pair from the other side gives the same bits (forward and reverse pass agree)
    rows = np.repeat(np.arange(500, 1500, dtype=np.int64)[:, None], k, axis=1)
    flat_y = ii.reshape(-1)
    order = np.argsort(flat_y, kind="stable")
    # query side = y rows (sorted so that a row range can be addressed), index side = q
    ysel, qsel = flat_y[order][:2000], rows.reshape(-1)[order][:2000]
    v_flat = v.reshape(-1)[order][:2000]
    for yrow in np.unique(ysel)[:50]:
        m = ysel == yrow
        idx = ctx.to_device(np.ascontiguousarray(qsel[m][None, :]))
        out2 = ctx.empty((1, int(m.sum())), np.float64)
        N._check(ctx.lib.kz_pair_values(ctx.handle, ym.handle, int(yrow), 1, qm.handle, idx.ptr, int(m.sum()), out2.ptr), "kz_pair_values")
        np.testing.assert_array_equal(out2.numpy()[0], v_flat[m])


@pytest.mark.parametrize("n,segs,L,k,with_dist,with_ind", [
    (5000, 8, 10, 10, True, True),        # the north-star exchange: 8 shards x K = 10
    (3000, 8, 50, 50, True, True),        # C3 on 8 GPUs: 400 entries per row
    (2000, 2, 7, 7, False, False),        # distance-only kinds
    (700, 8, 128, 128, True, True),       # 1024 entries per row
    (300, 3, 17, 40, True, False),        # k > segment length
    (64, 64, 100, 100, True, True),       # 6400 entries per row (one wave per workgroup)
    (1, 1, 5, 3, True, True),
])
def test_merge_topk_against_a_lexicographic_sort(ctx, n, segs, L, k, with_dist, with_ind):
    from kiez_amd import _native as N
    rng = np.random.default_rng(n + segs)
    M = segs * L
    key = np.round(rng.random((n, M)) * 40) / 40 if L % 2 == 0 else rng.random((n, M))    # even L: plenty of exact ties
    ind = rng.integers(0, 1 << 40, (n, M)).astype(np.int64)
    if L % 2 == 0:
        ind = (ind % 7).astype(np.int64)                                                   # ... also in (key, ind)
    for s in range(segs):   # the contract: every segment sorted by (key, ind)
        o = np.lexsort((ind[:, s * L:(s + 1) * L], key[:, s * L:(s + 1) * L]), axis=1)
        key[:, s * L:(s + 1) * L] = np.take_along_axis(key[:, s * L:(s + 1) * L], o, axis=1)
        ind[:, s * L:(s + 1) * L] = np.take_along_axis(ind[:, s * L:(s + 1) * L], o, axis=1)
    dist = key * 3.0 + 1.0
    dk, di, dd = ctx.to_device(key), ctx.to_device(ind), ctx.to_device(dist)
    od, oi = ctx.empty((n, k), np.float64), ctx.empty((n, k), np.int64)
    N._check(ctx.lib.kz_merge_topk(ctx.handle, dk.ptr, di.ptr if with_ind else None, dd.ptr if with_dist else None, n, segs, L, k,
                                   od.ptr, oi.ptr), "kz_merge_topk")
    ii = ind if with_ind else np.zeros_like(ind)
    order = np.lexsort((ii, key), axis=1)[:, :k]          # stable: ties fall back to the column = (segment, position)
    np.testing.assert_array_equal(oi.numpy(), np.take_along_axis(ii, order, axis=1))
    np.testing.assert_array_equal(od.numpy(), np.take_along_axis(dist if with_dist else key, order, axis=1))


def test_merge_topk_rejects_bad_shapes(ctx):
    from kiez_amd import _native as N
    a = ctx.to_device(np.zeros((4, 20)))
    od, oi = ctx.empty((4, 30), np.float64), ctx.empty((4, 30), np.int64)
    with pytest.raises(ValueError):
        N._check(ctx.lib.kz_merge_topk(ctx.handle, a.ptr, None, None, 4, 2, 10, 21, od.ptr, oi.ptr), "kz_merge_topk")
    with pytest.raises(ValueError):
        N._check(ctx.lib.kz_merge_topk(ctx.handle, a.ptr, None, None, 4, 100, 100, 5, od.ptr, oi.ptr), "kz_merge_topk")


@pytest.mark.parametrize("n,K,k", [(1000, 129, 129), (777, 200, 50), (300, 1000, 1000), (65, 1500, 3), (40, 4096, 100), (500, 256, 1)])
def test_wide_select_topk_is_the_reference_selection_sort(ctx, n, K, k):
    """HubnessReduction._sort (base.py:72-87) for more than 128 candidates: tie-heavy rows, NaN last."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(K)
    dist = np.round(rng.random((n, K)) * 20) / 20       # multiples of 1/20: every row is full of ties
    dist[::7, 5] = np.nan
    dist[3] = np.nan
    ind = np.argsort(rng.random((n, K)), axis=1).astype(np.int64)
    od, oi = N.select_topk(ctx, ctx.to_device(dist), ctx.to_device(ind), k)
    if k >= 2:    # (k = 1: numpy's SIMD arg-select is not the first minimum; the scalar rule is what we implement)
        # the reference's own expression (base.py:81-86); the oracle's restatement is checked against it on the NaN-free rows
        o = np.argpartition(dist, kth=np.arange(k), axis=1)[:, :k]
        rd, ri = np.take_along_axis(dist, o, axis=1), np.take_along_axis(ind, o, axis=1)
        clean = ~np.isnan(dist).any(axis=1)
        qd, qi = O.sort_topk(dist[clean], ind[clean], k)
        np.testing.assert_array_equal(qi, ri[clean])
    else:
        first = np.array([np.flatnonzero(np.isnan(r))[0] if np.isnan(r).all() else np.nanargmin(r) for r in dist])
        rd, ri = dist[np.arange(n), first][:, None], ind[np.arange(n), first][:, None]
    np.testing.assert_array_equal(oi.numpy(), ri)
    np.testing.assert_array_equal(od.numpy(), rd)


@pytest.mark.parametrize("K", [150, 300])
def test_more_than_128_candidates_through_the_api(K):
    """n_candidates > 128 through Kiez for every hubness kind against the oracle pipeline (the reference accepts any
    n_candidates > 1, base.py:20-27)."""
    from kiez_amd import Kiez
    from oracle import kiez_oracle as O
    from tests.golden_util import knife_edge_rows, knife_edge_topk_ok
    rng = np.random.RandomState(K)
    s, t = (rng.rand(700, 24), rng.rand(900, 24)) if K <= 150 else (rng.rand(360, 24), rng.rand(440, 24))   # (the oracle's MP-empiric loop is O(rows K^2))
    kinds = ((None, {}, "euclidean"), ("CSLS", {}, "euclidean"), ("LocalScaling", {"method": "standard"}, "euclidean"),
             ("LocalScaling", {"method": "nicdm"}, "cosine"), ("MutualProximity", {"method": "normal"}, "euclidean"),
             ("MutualProximity", {"method": "empiric"}, "euclidean"), ("DisSimLocal", {}, "sqeuclidean"))
    if K > 150:      # (the wave-per-row builds of every transform are the same code at K = 150 and 300: the larger K on the kinds whose kernels differ most)
        kinds = tuple(kd for kd in kinds if kd[0] in (None, "CSLS", "MutualProximity", "DisSimLocal"))
    for hub, kw, metric in kinds:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kz = Kiez(n_candidates=K, algorithm="SklearnNN", algorithm_kwargs={"metric": metric}, hubness=hub, hubness_kwargs=dict(kw))
            d, i = kz.fit(s, t).kneighbors(K - 20)
            od, oi = O.kiez_pipeline(s, t, K, K - 20, metric, 2, hub, kw)
        keep = np.ones(len(i), dtype=bool)
        if hub == "MutualProximity" and kw["method"] == "empiric":
            keep = ~knife_edge_rows(O.knn_exact(s, t, K, metric)[1])
            ri = O.knn_exact(t, s, K, metric)[1]
            for r in np.flatnonzero(~keep):
                assert knife_edge_topk_ok(od[r], oi[r], d[r], i[r], r, K, ri), (hub, r)
        np.testing.assert_array_equal(i[keep], oi[keep], err_msg=str((hub, kw)))
        np.testing.assert_allclose(d[keep], od[keep], rtol=1e-5, atol=1e-6 if hub == "DisSimLocal" else 1e-9, err_msg=str((hub, kw)))


def test_rows_only_matrix_is_a_row_source_and_nothing_else():
    """kz_matrix_create rows_on_device = 3: the gathered source of the multi-rank DSL path -- rows for kz_dsl_fit's centroid gather
    (dis_sim.py:96-101), no norms, no operand images; every search entry point refuses it."""
    import ctypes as C
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    ctx = N.Context.get()
    rng = np.random.RandomState(4)
    s, t = rng.rand(3000, 40).astype(np.float32), rng.rand(2500, 40).astype(np.float32)
    s_dev = ctx.to_device(s)
    sm_rows = N.DeviceMatrix(ctx, None, "euclidean", device_ptr=s_dev.ptr.value, shape=s.shape, dtype=np.float32, borrow=True,
                             keepalive=s_dev, rows_only=True)
    sm, tm = N.DeviceMatrix(ctx, s, "euclidean"), N.DeviceMatrix(ctx, t, "euclidean")
    _, i_t2s, _ = N.knn(ctx, tm, sm, 10)
    out_a, out_b = ctx.empty((len(t),), np.float64), ctx.empty((len(t),), np.float64)
    for src, out in ((sm, out_a), (sm_rows, out_b)):
        N._check(ctx.lib.kz_dsl_fit(ctx.handle, i_t2s.ptr, len(t), 10, src.handle, tm.handle, 0, out.ptr), "kz_dsl_fit")
    np.testing.assert_array_equal(out_a.numpy(), out_b.numpy())
    np.testing.assert_allclose(out_b.numpy(), O.dsl_fit(i_t2s.numpy(), s.astype(np.float64), t.astype(np.float64)), rtol=1e-12)
    with pytest.raises(ValueError, match="rows-only"):
        N.knn(ctx, tm, sm_rows, 5)
    with pytest.raises(ValueError, match="rows-only"):
        N.knn_dual(ctx, sm_rows, tm, 5)
    with pytest.raises(ValueError):
        N.DeviceMatrix(ctx, None, "euclidean", device_ptr=s_dev.ptr.value, shape=s.shape, dtype=np.float32, borrow=False, rows_only=True)
