"""The fp16 kernel's WIDE builds (context option "h_wide" = 1: one workgroup of 8 or 12 waves per CU whose query tiles share
one LDS ring, kz_knn_h16.h) must give, bit for bit, what the narrow builds give -- ordinary search and shared sweep, query
tile counts that do not divide by the tiles per workgroup, both occupancy classes.  Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    c = N.Context.get()
    yield c
    for name, value in (("h_wide", 0), ("dual_force", 0), ("chunk_rows", 0)):
        c.set_option(name, value)


@pytest.mark.parametrize("n_q,n_i,d,k", [
    (1000, 40000, 200, 10),      # 8 query tiles: units of three, the last workgroup reaches past the end (13 slices, 12 waves)
    (128 * 7 + 5, 30000, 300, 5),  # 8 tiles, the last one ragged; 19 slices: two tiles per workgroup (8 waves)
    (128 * 3, 20000, 144, 10),   # exactly one wide workgroup; 9 slices
    (50, 9000, 208, 3),          # a single query tile: two of three wave groups idle
])
def test_wide_builds_equal_the_narrow_ones(ctx, n_q, n_i, d, k):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(n_q + d)
    q = rng.random((n_q, d), dtype=np.float32)
    y = rng.random((n_i, d), dtype=np.float32)
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    res = {}
    for wide in (0, 1):
        ctx.set_option("h_wide", wide)
        dd, ii, st = N.knn(ctx, qm, ym, k)
        assert st["max_err_ratio"] < 1.0 and st["first_pass"] == 2, st
        res[wide] = (dd.numpy(), ii.numpy(), st["n_blocks"])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][0], res[1][0])
    assert res[1][2] < res[0][2] or n_q <= 128, (res[0][2], res[1][2])   # fewer, wider workgroups actually ran
    od, oi = O.knn_exact(q[:200], y, k, "euclidean")
    np.testing.assert_array_equal(res[1][1][:200], oi)
    np.testing.assert_array_equal(res[1][0][:200], od)


@pytest.mark.parametrize("na,nb,d,k", [(20000, 7000, 200, 10), (9000, 12000, 300, 10)])
def test_wide_shared_sweep_equals_two_searches(ctx, na, nb, d, k):
    from kiez_amd import _native as N
    rng = np.random.default_rng(na)
    a = rng.random((na, d), dtype=np.float32)
    b = rng.random((nb, d), dtype=np.float32)
    am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
    ctx.set_option("h_wide", 0)
    d_ab, i_ab, _ = N.knn(ctx, am, bm, k)
    d_ba, i_ba, _ = N.knn(ctx, bm, am, k)
    ctx.set_option("h_wide", 1)
    ctx.set_option("dual_force", 1)
    (xd, xi, s_ab), (yd, yi, s_ba) = N.knn_dual(ctx, am, bm, k)
    assert s_ab["dual"] == 1 and s_ba["dual"] == 1, (s_ab, s_ba)
    np.testing.assert_array_equal(xi.numpy(), i_ab.numpy())
    np.testing.assert_array_equal(xd.numpy(), d_ab.numpy())
    np.testing.assert_array_equal(yi.numpy(), i_ba.numpy())
    np.testing.assert_array_equal(yd.numpy(), d_ba.numpy())
