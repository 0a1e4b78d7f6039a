"""The range re-search (kz_range.h): rows that no approximate tier certifies go to the exact float64 kernels on the index rows whose
approximate key lies within the rounding bound of the row's k-th best candidate -- not on the whole index.  Same neighbours and the
same distance bits as the whole-index kernels (option exact_rows = 2) on every metric of the inner-product family, on partial ranges
(an inflated rounding bound: every row fails every tier, the range holds part of the index), full ranges, duplicates, a zero row,
exclude_self, and on the two hand-back paths (a log that overflows; rows without k candidates).  The reference has one brute-force
search for all of it: kiez/neighbors/exact/sklearn_nearest_neighbors.py:96-101.  `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    c = N.Context.get()
    yield c
    c.set_option("eps_scale", 1.0)
    c.set_option("exact_rows", 3)
    c.set_option("abl", 0)
    c.set_option("spec_rows", 64)
    c.set_option("chunk_rows", 0)


def _tight(n, d, seed, n_clusters=6):
    """clusters three orders of magnitude tighter than their distance from the centre, rows shuffled (tools/cliff_probe.py)"""
    rng = np.random.default_rng(seed)
    centres = np.random.default_rng(5).standard_normal((n_clusters, d)) * 3
    sc = 0.01 * 2.0 ** np.random.default_rng(6).integers(0, 4, n_clusters)
    c = rng.integers(0, n_clusters, n)
    return (centres[c] + sc[c, None] * rng.standard_normal((n, d))).astype(np.float32)


def _both(ctx, qm, ym, k, **kw):
    from kiez_amd import _native as N
    out = {}
    for rows in (3, 2):
        ctx.set_option("exact_rows", rows)
        dd, ii, st = N.knn(ctx, qm, ym, k, **kw)
        out[rows] = (dd.numpy(), ii.numpy(), st)
    ctx.set_option("exact_rows", 3)
    np.testing.assert_array_equal(out[3][1], out[2][1])
    np.testing.assert_array_equal(out[3][0], out[2][0])
    assert out[2][2]["n_range_rows"] == 0
    return out[3]


@pytest.mark.parametrize("metric", ["euclidean", "sqeuclidean", "cosine"])
@pytest.mark.parametrize("d", [32, 64, 200, 300])      # (from 32 elements on: the fp16 image needs two 16-element slices)
def test_tight_clusters_same_bits_and_oracle(ctx, metric, d):
    """Data the tiers cannot certify by themselves: the tightest clusters' rows end on the range re-search."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    q, y = _tight(3000, d, 1), _tight(9001, d, 2)
    y[5] = y[17]
    qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
    dd, ii, st = _both(ctx, qm, ym, 10)
    assert st["n_range_rows"] <= st["n_fallback_rows"], st
    if metric != "cosine":      # (normalised, the clusters are tighter still: few rows get past the float32-operand tier)
        assert st["n_range_rows"] >= 128, st
        assert 0 < st["n_range_pairs"] < st["n_range_rows"] * y.shape[0], st        # (a part of the index, not all of it)
    q64, y64 = (q.astype(np.float64), y.astype(np.float64)) if metric == "cosine" else (q, y)
    od, oi = O.knn_exact(q64, y64, 10, metric)
    np.testing.assert_array_equal(ii, oi)
    np.testing.assert_allclose(dd, od, rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("eps_scale", [30.0, 1000.0, 1e30])
@pytest.mark.parametrize("metric,d,k", [("euclidean", 32, 10), ("cosine", 100, 50), ("sqeuclidean", 260, 5)])
def test_inflated_bound_partial_and_full_ranges(ctx, eps_scale, metric, d, k):
    """An inflated rounding bound on gaussian rows: every row fails every tier; the range is a part of the index (30 x), most of it
    (1000 x) or all of it (1e30: the thresholds are -inf)."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(d)
    q = rng.standard_normal((333, d)).astype(np.float32)
    y = rng.standard_normal((7003, d)).astype(np.float32)
    y[5] = y[17]
    y[100] = 0.0
    q[3] = y[5]
    qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
    ctx.set_option("eps_scale", eps_scale)
    dd, ii, st = _both(ctx, qm, ym, k)
    if eps_scale == 1e30:
        assert st["n_fallback_rows"] == 333 and st["n_range_rows"] == 333 and st["n_range_pairs"] == 333 * 7003, st
    else:
        assert st["n_range_pairs"] < 333 * 7003, st
    q64, y64 = (q.astype(np.float64), y.astype(np.float64)) if metric == "cosine" else (q, y)
    od, oi = O.knn_exact(q64, y64, k, metric)
    np.testing.assert_array_equal(ii, oi)


def test_exclude_self(ctx):
    from kiez_amd import _native as N
    y = _tight(6000, 48, 3)
    ym = N.DeviceMatrix(ctx, y, "euclidean")
    dd, ii, st = _both(ctx, ym, ym, 10, exclude_self=True)
    assert st["n_range_rows"] >= 128, st
    assert not (ii == np.arange(6000)[:, None]).any()


@pytest.mark.parametrize("abl", [4, 8])
def test_rows_handed_back(ctx, abl):
    """abl 4: the log overflows at every batch size -- the batch goes back to the whole-index kernels; abl 8: no row has a bound --
    every segment is empty, every row goes back.  Same results either way."""
    from kiez_amd import _native as N
    q, y = _tight(3000, 64, 1), _tight(9001, 64, 2)
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    dd0, ii0, st0 = N.knn(ctx, qm, ym, 10)
    assert st0["n_range_rows"] >= 128
    ctx.set_option("abl", abl)
    dd1, ii1, st1 = N.knn(ctx, qm, ym, 10)
    ctx.set_option("abl", 0)
    assert st1["n_range_rows"] == 0 and st1["n_fallback_rows"] > 0, (st0, st1)      # (the rows went on: next tier, whole-index kernels)
    np.testing.assert_array_equal(ii0.numpy(), ii1.numpy())
    np.testing.assert_array_equal(dd0.numpy(), dd1.numpy())


@pytest.mark.parametrize("metric,d,k", [("euclidean", 64, 10), ("cosine", 200, 50), ("sqeuclidean", 300, 5)])
def test_groups_same_bits(ctx, metric, d, k):
    """Thousands of uncertified rows: rows of one tight cluster share a representative's range (a dense block per group), tried on
    what the split-bf16 pass leaves and again at the end of the ladder.  Three routes, the same bits: groups (default), one range per
    row (abl 16), the whole index (exact_rows 2)."""
    from kiez_amd import _native as N
    q, y = _tight(14000, d, 11, n_clusters=5), _tight(20001, d, 12, n_clusters=5)
    y[5] = y[17]
    qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
    dd, ii, st = _both(ctx, qm, ym, k)
    assert st["n_range_group_rows"] >= 2048 and st["n_range_group_rows"] <= st["n_range_rows"] <= st["n_fallback_rows"], st
    assert st["n_range_pairs"] < st["n_range_rows"] * y.shape[0] // 2, st
    ctx.set_option("abl", 16)
    d1, i1, s1 = N.knn(ctx, qm, ym, k)
    ctx.set_option("abl", 0)
    assert s1["n_range_group_rows"] == 0, s1
    np.testing.assert_array_equal(ii, i1.numpy())
    np.testing.assert_array_equal(dd, d1.numpy())


def test_small_chunks(ctx):
    """The test knob chunk_rows (query rows per launch): the range sweeps take batches no larger than a chunk -- same results."""
    from kiez_amd import _native as N
    q, y = _tight(3000, 64, 1), _tight(9001, 64, 2)
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    d0, i0, st0 = N.knn(ctx, qm, ym, 10)
    ctx.set_option("chunk_rows", 512)
    d1, i1, st1 = N.knn(ctx, qm, ym, 10)
    ctx.set_option("chunk_rows", 0)
    assert st1["n_range_rows"] > 0, st1
    np.testing.assert_array_equal(i0.numpy(), i1.numpy())
    np.testing.assert_array_equal(d0.numpy(), d1.numpy())


def test_both_directions_of_a_fit(ctx):
    """kz_knn_dual on tight clusters: both directions' uncertified rows take the range re-search."""
    from kiez_amd import _native as N
    a, b = _tight(5000, 64, 4), _tight(5200, 64, 5)
    am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
    (d1, i1, s1), (d2, i2, s2) = N.knn_dual(ctx, am, bm, 10)
    ctx.set_option("exact_rows", 2)
    (e1, j1, t1), (e2, j2, t2) = N.knn_dual(ctx, am, bm, 10)
    ctx.set_option("exact_rows", 3)
    assert s1["n_range_rows"] + s2["n_range_rows"] > 0 and t1["n_range_rows"] + t2["n_range_rows"] == 0
    for x, z in ((d1, e1), (i1, j1), (d2, e2), (i2, j2)):
        np.testing.assert_array_equal(x.numpy(), z.numpy())
