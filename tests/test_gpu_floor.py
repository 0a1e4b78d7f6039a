"""Seeded lists ("population floor", kiez_amd/csrc/kz_knn.hip): the candidate lists of a large sweep start at a per-row floor
taken from a probe of the query rows instead of at -inf.  The floor may only decide how many rows are searched again -- never a
result: with the floor, without it, and with a floor that is deliberately far too high, indices and distances are identical,
bit for bit, in the shared sweep (both directions) and in an ordinary search."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_RESET = (("list_floor", 1), ("floor_margin", 1.3), ("dual_force", 0), ("tier_probe", 1024), ("h_q64", 2))


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    c = N.Context.get()
    yield c
    for name, value in _RESET:
        c.set_option(name, value)


def _data(kind, n, d, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, d)).astype(np.float32)
    if kind == "normal":
        return rng.standard_normal((n, d)).astype(np.float32)
    # clusters of very different density, rows in cluster order: the k-th key is anything but a function of |q_c|^2
    centres = rng.standard_normal((12, d)) * 2
    scale = 0.05 + rng.random(12)
    which = np.sort(rng.integers(0, 12, n))
    return (centres[which] + scale[which, None] * rng.standard_normal((n, d))).astype(np.float32)


def _dual(ctx, a, b, k, metric):
    from kiez_amd import _native as N
    am, bm = N.DeviceMatrix(ctx, a, metric), N.DeviceMatrix(ctx, b, metric)
    (xd, xi, s_ab), (yd, yi, s_ba) = N.knn_dual(ctx, am, bm, k)
    return (xd.numpy(), xi.numpy(), yd.numpy(), yi.numpy()), s_ab, s_ba


@pytest.mark.parametrize("kind,metric,k,d,q64", [("uniform", "euclidean", 10, 72, 0), ("uniform", "cosine", 50, 200, 0), ("normal", "sqeuclidean", 10, 200, 1),
                                                 ("clustered", "euclidean", 10, 48, 0), ("clustered", "cosine", 30, 64, 1)])
def test_shared_sweep_results_do_not_depend_on_the_floor(ctx, kind, metric, k, d, q64):
    a, b = _data(kind, 20_000, d, 1), _data(kind, 33_000, d, 2)
    ctx.set_option("dual_force", 1)
    ctx.set_option("h_q64", q64)
    outs, stats = [], []
    for floor, margin in ((0, 1.3), (1, 1.3), (1, 0.0), (1, 50.0)):
        ctx.set_option("list_floor", floor)
        ctx.set_option("floor_margin", margin)
        o, s_ab, s_ba = _dual(ctx, a, b, k, metric)
        assert s_ab["dual"] == 1
        outs.append(o)
        stats.append(s_ab)
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            np.testing.assert_array_equal(x.view(np.int64), y.view(np.int64))
    # margin 0: the floor is the model itself -- about half of the rows end short of k candidates and are searched again
    assert stats[2]["n_escalated_rows"] > stats[0]["n_escalated_rows"] + 2_000
    # the default margin sends at most a few rows more down that road than no floor at all (~n / probe rows expected)
    assert stats[1]["n_escalated_rows"] <= stats[0]["n_escalated_rows"] + 400


def test_ordinary_search_results_do_not_depend_on_the_floor(ctx):
    from kiez_amd import _native as N
    q, y = _data("uniform", 231_000, 32, 3), _data("uniform", 231_000, 32, 4)   # (5.3e10 pairs: the tier probe's gate)
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    outs, stats = [], []
    for floor, margin in ((0, 1.3), (1, 1.3), (1, 0.0)):
        ctx.set_option("list_floor", floor)
        ctx.set_option("floor_margin", margin)
        d, i, st = N.knn(ctx, qm, ym, 10)
        outs.append((d.numpy(), i.numpy()))
        stats.append(st)
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0][1], o[1])
        np.testing.assert_array_equal(outs[0][0].view(np.int64), o[0].view(np.int64))
    assert stats[2]["n_escalated_rows"] > 50_000
    assert stats[1]["n_escalated_rows"] <= stats[0]["n_escalated_rows"] + 600
    # ... and a sample of rows against the oracle
    from oracle import kiez_oracle as O
    rows = np.random.default_rng(0).choice(len(q), 64, replace=False)
    od, oi = O.knn_exact(q[rows], y, 10, "euclidean")
    np.testing.assert_array_equal(outs[1][1][rows], oi)


def test_exclude_self_with_a_floor(ctx):
    from kiez_amd import _native as N
    x = _data("uniform", 231_000, 32, 7)
    xm = N.DeviceMatrix(ctx, x, "euclidean")
    outs = []
    for floor in (0, 1):
        ctx.set_option("list_floor", floor)
        d, i, st = N.knn(ctx, xm, xm, 10, exclude_self=True)
        outs.append((d.numpy(), i.numpy()))
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][0].view(np.int64), outs[1][0].view(np.int64))
    assert not (outs[1][1] == np.arange(len(x))[:, None]).any()
