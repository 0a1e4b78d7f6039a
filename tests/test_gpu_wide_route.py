"""WIDE ROUTE of the fp16 tier (kz_knn.hip "WIDE ROUTE"; round 5): data whose keys are DENSE around the k-th neighbour -- tight
clusters: hundreds of rows within the rounding bound of the k-th key -- needs margin in RANKS, not better operands.  The tier
probe's ladder finds that out (default lists -> 32 lists of 16 on the same fp16 operands -> split-bf16) and the call keeps the
one-product-per-multiply-add kernel.  Whatever the route, the result is the reference's float64 neighbour order
(sklearn_nearest_neighbors.py:96-101).  Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OPTS = (("tier_probe", 1024), ("probe_min_pairs", 5e10), ("wide_lists", 32), ("wide_sel", 256), ("dual_force", 0), ("eps_scale", 1.0), ("esc_ladder", 1),
        ("probe_min_pairs", 5e10))


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    c = N.Context.get()
    yield c
    for name, value in OPTS:
        c.set_option(name, value)


def clustered(n, d, seed, clusters=12, spread=0.4, far=3.0, shuffle=False):
    """Tight gaussian clusters far from the centre (bench.py "hard"): the same centres for every seed."""
    centres = np.random.RandomState(5).standard_normal((clusters, d)) * far
    rng = np.random.RandomState(seed)
    sizes = rng.multinomial(n, np.ones(clusters) / clusters)
    x = np.concatenate([centres[c] + spread * rng.standard_normal((sizes[c], d)) for c in range(clusters)]).astype(np.float32)
    if shuffle:
        x = x[rng.permutation(n)]
    return x


@pytest.mark.parametrize("metric,k,shuffle", [("cosine", 50, False), ("euclidean", 10, True), ("sqeuclidean", 30, False)])
def test_probe_ladder_takes_the_wide_route_and_every_row_is_the_oracles(ctx, metric, k, shuffle):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    q, y = clustered(20_000, 64, 1, shuffle=shuffle), clustered(60_000, 64, 2, shuffle=shuffle)
    qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
    ctx.set_option("probe_min_pairs", 0)
    ctx.set_option("tier_probe", 1024)
    d1, i1, s1 = N.knn(ctx, qm, ym, k)
    # the ladder chose the wide fp16 route (first pass: fp16 = 2), 32 lists of 16; what it left uncertified went down the tiers
    # (... or its first rung certified more than half of the probe and the call kept its ordinary lists: the k = 30 case)
    assert s1["first_pass"] == 2, s1
    if s1["wide_lists"]:
        assert s1["wide_lists"] == 32 and s1["list_len"] == 16 and s1["n_first_pass_fail"] <= len(q) // 3, s1
    else:
        assert k == 30 and s1["n_first_pass_fail"] <= len(q) * 6 // 10, s1
    ctx.set_option("wide_lists", 0)       # the ladder switched off: the same data starts at the split-bf16 tier
    d0, i0, s0 = N.knn(ctx, qm, ym, k)
    assert s0["wide_lists"] == 0 and s0["first_pass"] == (1 if s1["wide_lists"] else 2), s0
    np.testing.assert_array_equal(i1.numpy(), i0.numpy())
    np.testing.assert_array_equal(d1.numpy(), d0.numpy())
    q64, y64 = (q.astype(np.float64), y.astype(np.float64)) if metric == "cosine" else (q, y)
    od, oi = O.knn_exact(q64, y64, k, metric)
    np.testing.assert_array_equal(i1.numpy(), oi)
    np.testing.assert_allclose(d1.numpy(), od, rtol=1e-5, atol=1e-6)     # (north_star: 1e-5 relative)
    assert s1["max_err_ratio"] < 1.0


def test_uniform_data_never_sees_the_ladder(ctx):
    from kiez_amd import _native as N
    rng = np.random.RandomState(0)
    q, y = rng.rand(20_000, 64).astype(np.float32), rng.rand(60_000, 64).astype(np.float32)
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    ctx.set_option("probe_min_pairs", 0)
    _, _, s = N.knn(ctx, qm, ym, 10)
    assert s["first_pass"] == 2 and s["wide_lists"] == 0 and s["n_first_pass_fail"] < 100, s


def test_shared_sweep_hands_hard_data_to_two_wide_searches(ctx):
    """kz_knn_dual's own probe runs the same ladder: both directions as ordinary searches on the wide route, results those of
    two plain searches."""
    from kiez_amd import _native as N
    a, b = clustered(140_000, 64, 3), clustered(60_000, 64, 4)      # (large enough for the shared sweep's cost model to take it)
    am, bm = N.DeviceMatrix(ctx, a, "cosine"), N.DeviceMatrix(ctx, b, "cosine")
    ctx.set_option("probe_min_pairs", 0)
    (d_ab, i_ab, s_ab), (d_ba, i_ba, s_ba) = N.knn_dual(ctx, am, bm, 50)
    assert s_ab["dual"] == 0 and s_ab["wide_lists"] == 32 and s_ba["wide_lists"] == 32 and s_ab["first_pass"] == 2, (s_ab, s_ba)
    ctx.set_option("wide_lists", 0)
    ctx.set_option("tier_probe", 0)
    d0, i0, _ = N.knn(ctx, am, bm, 50)
    d1, i1, _ = N.knn(ctx, bm, am, 50)
    np.testing.assert_array_equal(i_ab.numpy(), i0.numpy())
    np.testing.assert_array_equal(d_ab.numpy(), d0.numpy())
    np.testing.assert_array_equal(i_ba.numpy(), i1.numpy())
    np.testing.assert_array_equal(d_ba.numpy(), d1.numpy())


@pytest.mark.parametrize("metric,k,n_q,n_i,shared", [("euclidean", 10, 30_000, 40_000, False), ("cosine", 50, 30_000, 31_000, True),
                                                     ("euclidean", 10, 9_000, 12_000, False)])
def test_ladder_after_the_fact_on_searches_too_small_for_a_probe(ctx, metric, k, n_q, n_i, shared):
    """Below the probe's size gates a pass finds out AFTERWARDS that fp16 with ordinary lists certifies next to nothing on this
    data; the failed rows then try the wide route on a sample of themselves before the split-bf16 tier (kz_knn.hip "LADDER AFTER
    THE FACT").  Results with and without that ladder are identical and the oracle's; with it the call is several times faster
    (tools/cliff_probe.py) -- here only: not slower by more than measurement noise."""
    import time
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    q, y = clustered(n_q, 64, 3, clusters=20, shuffle=True), clustered(n_i, 64, 4, clusters=20, shuffle=True)
    qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
    out, ms = {}, {}
    for ladder in (0, 1, 0, 1):
        ctx.set_option("esc_ladder", ladder)
        ctx.sync()
        t0 = time.perf_counter()
        if shared:
            (d, i, st), (e, j, st2) = N.knn_dual(ctx, qm, ym, k)
            res = (d.numpy(), i.numpy(), e.numpy(), j.numpy())
        else:
            d, i, st = N.knn(ctx, qm, ym, k)
            res = (d.numpy(), i.numpy())
        ctx.sync()
        ms[ladder] = (time.perf_counter() - t0) * 1e3
        assert st["n_escalated_rows"] >= 4096, st      # (the situation this test is about: most rows left the first pass uncertified)
        out[ladder] = res
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)
    rows = np.arange(0, n_q, max(n_q // 300, 1))
    q64, y64 = (q.astype(np.float64), y.astype(np.float64)) if metric == "cosine" else (q, y)
    od, oi = O.knn_exact(q64[rows], y64, k, metric)
    np.testing.assert_array_equal(out[1][1][rows], oi)
    assert ms[1] <= 1.25 * ms[0] + 1.0, ms
