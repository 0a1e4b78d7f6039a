"""Short-list route of the ORDINARY kernel (kz_knn.hip "SHORT-LIST ROUTE of the ordinary kernel"): 13 .. 110 neighbours per query as
lists of 16 over P index ranges of a second, row-dealt image of the index, instead of one list of 32 / 64 / 128.  The reference
has no counterpart (scikit-learn's brute force keeps one heap per query, sklearn_nearest_neighbors.py:96-101): the result must be
the float64 neighbour order all the same.  Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    c = N.Context.get()
    yield c
    for name, value in (("short_ord", 1), ("short_ord_min_tiles", 48), ("eps_scale", 1.0)):
        c.set_option(name, value)


def _data(kind, n, d, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, d)).astype(np.float32)
    if kind == "cluster by cluster":   # all near rows of a query in ONE stretch of the matrix
        centres = rng.standard_normal((12, d))
        sizes = rng.multinomial(n, np.ones(12) / 12)
        return np.concatenate([centres[c] + 0.5 * rng.standard_normal((sizes[c], d)) for c in range(12)]).astype(np.float32)
    if kind == "duplicates":
        base = rng.random((max(n // 9, 8), d))
        return base[rng.integers(0, len(base), n)].astype(np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind,metric", [("uniform", "euclidean"), ("cluster by cluster", "cosine"), ("duplicates", "sqeuclidean")])
@pytest.mark.parametrize("k", [13, 26, 50, 64, 80, 100, 110])
def test_route_on_and_off_give_the_same_neighbours_and_the_oracle_agrees(ctx, kind, metric, k):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    q, y = _data(kind, 3000, 48, 1), _data(kind, 30000, 48, 2)
    qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
    ctx.set_option("short_ord_min_tiles", 8)    # (235 index tiles: the route is taken with ranges of >= 8 tiles)
    ctx.set_option("short_ord", 0)
    d0, i0, s0 = N.knn(ctx, qm, ym, k)
    ctx.set_option("short_ord", 1)
    d1, i1, s1 = N.knn(ctx, qm, ym, k)
    assert s0["list_len"] in (32, 64, 128) and s1["list_len"] == 16 and s1["n_splits"] >= max(2, (k + 4) // 5 - 1), (s0, s1)   # (k = 110: 22 lists, 352 entries)
    np.testing.assert_array_equal(i1.numpy(), i0.numpy())
    np.testing.assert_array_equal(d1.numpy(), d0.numpy())
    if kind != "duplicates":
        rows = np.arange(0, len(q), 15)
        q64, y64 = (q.astype(np.float64), y.astype(np.float64)) if metric == "cosine" else (q, y)
        np.testing.assert_array_equal(i1.numpy()[rows], O.knn_exact(q64[rows], y64, k, metric)[1])


@pytest.mark.parametrize("k", [20, 50])
def test_route_in_single_source_mode_strips_the_query_itself(ctx, k):
    """exclude_self: the query's own row is a row of the DEALT image under another number -- the finalize kernel translates the
    list entries back before it strips it (neighbor_algorithm_base.py:119 is_self_querying)."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    y = _data("cluster by cluster", 20000, 40, 3)
    ym = N.DeviceMatrix(ctx, y, "euclidean")
    ctx.set_option("short_ord_min_tiles", 8)
    d1, i1, s1 = N.knn(ctx, ym, ym, k, exclude_self=True)
    assert s1["list_len"] == 16
    i = i1.numpy()
    assert (i != np.arange(len(y))[:, None]).all()
    rows = np.arange(0, len(y), 97)
    np.testing.assert_array_equal(i[rows], O.knn_exact(y, y, k, "euclidean", exclude_self=True)[1][rows])


def test_rows_the_route_cannot_certify_go_down_and_come_back_right(ctx):
    """eps_scale large enough that many rows fail the fp16 certification: they are searched again with LONG lists (never the
    same geometry twice), then with better operands; same neighbours as without the route."""
    from kiez_amd import _native as N
    q, y = _data("uniform", 2000, 32, 5), _data("uniform", 40000, 32, 6)
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    ctx.set_option("short_ord_min_tiles", 8)
    out = {}
    for short in (0, 1):
        ctx.set_option("short_ord", short)
        res = []
        for eps in (1.0, 40.0, 1e9):
            ctx.set_option("eps_scale", eps)
            d, i, st = N.knn(ctx, qm, ym, 30)
            res.append((i.numpy(), d.numpy(), st["n_escalated_rows"] + st["n_fallback_rows"]))
        ctx.set_option("eps_scale", 1.0)
        out[short] = res
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    assert out[1][2][2] >= 2000     # eps_scale = 1e9: every row reaches the exact float64 kernels
    for r in out[1][1:]:
        np.testing.assert_array_equal(r[0], out[1][0][0])


@pytest.mark.parametrize("k", [111, 128, 160, 200, 320])
def test_long_k_route_with_lists_of_sixteen(ctx, k):
    """111 .. 320 neighbours on an index large enough: k / 5 lists of 16 (up to 64: 1024 entries per query) instead of the long-k
    route's lists of 128; same neighbours either way, and the oracle's."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    q, y = _data("cluster by cluster", 1500, 32, 11), _data("cluster by cluster", 40000, 32, 12)
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    ctx.set_option("short_ord_min_tiles", 4)
    ctx.set_option("short_ord", 0)
    d0, i0, s0 = N.knn(ctx, qm, ym, k)
    ctx.set_option("short_ord", 1)
    d1, i1, s1 = N.knn(ctx, qm, ym, k)
    assert s0["list_len"] == 128 and s1["list_len"] == 16 and s1["n_splits"] >= (k + 4) // 5 - 1, (s0, s1)
    np.testing.assert_array_equal(i1.numpy(), i0.numpy())
    np.testing.assert_array_equal(d1.numpy(), d0.numpy())
    rows = np.arange(0, len(q), 25)
    np.testing.assert_array_equal(i1.numpy()[rows], O.knn_exact(q[rows], y, k, "euclidean")[1])


def test_tier_probe_starts_hard_data_at_the_split_bf16_tier():
    """Data that is hard for fp16 as a whole (tight clusters far from the centre: every row fails the first pass' certification): a
    strided 4096-row probe finds that out and the search starts at the split-bf16 tier instead of paying for a whole fp16 sweep;
    easy data of the same size keeps the fp16 pass.  Identical results either way (sklearn_nearest_neighbors.py:96-101)."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    ctx = N.Context.get()
    rng = np.random.RandomState(5)
    d, n_q, n_i = 64, 300_000, 200_000
    centres = rng.standard_normal((40, d)) * 3

    def clustered(n):
        sizes = rng.multinomial(n, np.ones(40) / 40)
        return np.concatenate([centres[c] + 0.4 * rng.standard_normal((sizes[c], d)) for c in range(40)]).astype(np.float32)
    try:
        for kind, q, y in (("hard", clustered(n_q), clustered(n_i)), ("easy", rng.rand(n_q, d).astype(np.float32), rng.rand(n_i, d).astype(np.float32))):
            qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
            ctx.set_option("tier_probe", 4096)
            ctx.set_option("wide_lists", 0)      # the ladder's second rung off: what the round-4 probe did
            d1, i1, s1 = N.knn(ctx, qm, ym, 10)
            ctx.set_option("wide_lists", 32)     # round 5: before better operands, more margin in ranks on the same ones
            d2, i2, s2 = N.knn(ctx, qm, ym, 10)
            ctx.set_option("tier_probe", 0)
            d0, i0, s0 = N.knn(ctx, qm, ym, 10)
            assert s0["first_pass"] == 2                                     # without the probe: always the fp16 pass first
            assert s1["first_pass"] == (1 if kind == "hard" else 2), (kind, s1)
            assert s2["first_pass"] == 2 and s2["wide_lists"] == (32 if kind == "hard" else 0), (kind, s2)
            for dd, ii in ((d1, i1), (d2, i2)):
                np.testing.assert_array_equal(ii.numpy(), i0.numpy())
                np.testing.assert_array_equal(dd.numpy(), d0.numpy())
            rows = np.arange(0, n_q, n_q // 300)[:300]
            od, oi = O.knn_exact(q[rows], y, 10, "euclidean")
            np.testing.assert_array_equal(i1.numpy()[rows], oi)
            if kind == "hard":
                assert s0["n_escalated_rows"] > n_q // 2
    finally:
        ctx.set_option("tier_probe", 1024)
        ctx.set_option("wide_lists", 32)


def test_shared_sweep_leaves_hard_data_to_two_searches():
    """The shared sweep looks first, too (its floor probe is its tier probe): on data that is hard for fp16 as a whole it hands the
    call to two ordinary searches, which take the route the probe's ladder found (the fp16 tier's wide route, else the split-bf16
    tier) -- instead of sweeping in fp16 and sending nearly every row of
    both directions down the tiers afterwards (bench.py "hard": 127 against 201 ms per step).  Easy data of the same size keeps the
    shared sweep.  Identical results either way."""
    from kiez_amd import _native as N
    ctx = N.Context.get()
    rng = np.random.RandomState(6)
    d, n_a, n_b = 64, 300_000, 200_000
    centres = rng.standard_normal((40, d)) * 3

    def clustered(n):
        sizes = rng.multinomial(n, np.ones(40) / 40)
        return np.concatenate([centres[c] + 0.4 * rng.standard_normal((sizes[c], d)) for c in range(40)]).astype(np.float32)
    try:
        for kind, a, b in (("hard", clustered(n_a), clustered(n_b)), ("easy", rng.rand(n_a, d).astype(np.float32), rng.rand(n_b, d).astype(np.float32))):
            am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
            ctx.set_option("tier_probe", 4096)
            (xd, xi, sa), (yd, yi, sb) = N.knn_dual(ctx, am, bm, 10)
            assert sa["dual"] == (0 if kind == "hard" else 1), (kind, sa)
            if kind == "hard":
                # both searches took what the probe's ladder found: the fp16 tier's wide route (round 5) -- or, where its second
                # rung fails too, the split-bf16 tier
                assert (sa["first_pass"], sa["wide_lists"]) in ((2, 32), (1, 0)) and (sb["first_pass"], sb["wide_lists"]) in ((2, 32), (1, 0)), (sa, sb)
                assert sa["wide_lists"] == 32                                     # (this set: the wide route)
            d_ab, i_ab, _ = N.knn(ctx, am, bm, 10)
            d_ba, i_ba, _ = N.knn(ctx, bm, am, 10)
            np.testing.assert_array_equal(xi.numpy(), i_ab.numpy())
            np.testing.assert_array_equal(xd.numpy(), d_ab.numpy())
            np.testing.assert_array_equal(yi.numpy(), i_ba.numpy())
            np.testing.assert_array_equal(yd.numpy(), d_ba.numpy())
    finally:
        ctx.set_option("tier_probe", 1024)
        ctx.set_option("wide_lists", 32)
