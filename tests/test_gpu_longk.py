"""More than 110 neighbours per query on the fused kernels (long-k route of kz_knn: lists of 128 over many index ranges, a
finalize kernel that selects k + margin candidates from their union): the reference's SklearnNN accepts any k <= n
(kiez/neighbors/exact/sklearn_nearest_neighbors.py:51-65, 96-101).  Against the oracle, every tier, including data whose
nearest rows all sit in ONE index range (the certification must notice and send those rows down).  `pytest -m gpu`."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    c = N.Context.get()
    yield c
    for name, value in (("precision", 0), ("eps_scale", 1.0)):
        c.set_option(name, value)


@pytest.mark.parametrize("n_q,n_i,d,k,metric,dtype,prec", [
    (3000, 40000, 64, 111, "euclidean", np.float32, 0),
    (2000, 60000, 200, 256, "cosine", np.float32, 0),
    (1500, 80000, 48, 512, "sqeuclidean", np.float64, 0),
    (1000, 50000, 128, 300, "euclidean", np.float32, 2),     # split-bf16 tier
    (800, 30000, 32, 200, "euclidean", np.float32, 1),       # float32-operand tier (two lane-half lists per range)
    (500, 70000, 100, 540, "euclidean", np.float32, 0),      # the largest k the route takes
])
def test_long_k_against_the_oracle(ctx, n_q, n_i, d, k, metric, dtype, prec):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(k)
    q = rng.random((n_q, d)).astype(dtype)
    y = rng.random((n_i, d)).astype(dtype)
    ctx.set_option("precision", prec)
    dd, ii, st = N.knn(ctx, N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric), k)
    assert st["n_splits"] >= 4 and st["list_len"] == 128 and st["main_kernel_ms"] > 0, st     # the fused kernels ran
    assert st["n_fallback_rows"] < n_q // 10 and st["max_err_ratio"] < 1.0, st
    q64, y64 = (q.astype(np.float64), y.astype(np.float64)) if metric == "cosine" else (q, y)
    od, oi = O.knn_exact(q64, y64, k, metric)
    np.testing.assert_array_equal(ii.numpy(), oi)
    np.testing.assert_allclose(dd.numpy(), od, rtol=1e-6 if metric == "cosine" else 1e-12, atol=1e-7 if metric == "cosine" else 1e-12)


def test_long_k_with_all_neighbours_in_one_index_range(ctx):
    """Index rows in cluster order: the 300 nearest rows of a query are consecutive rows -- far more than one range's list
    holds.  The certification has to see it (full list whose smallest key beats the selection) and the rows must still
    come out exact (down the tiers to the exact kernels)."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(5)
    centres = rng.standard_normal((40, 24)) * 6
    y = np.concatenate([c + 0.1 * rng.standard_normal((1000, 24)) for c in centres]).astype(np.float32)   # 40 blocks of 1000
    q = (centres[rng.integers(0, 40, 300)] + 0.1 * rng.standard_normal((300, 24))).astype(np.float32)
    dd, ii, st = N.knn(ctx, N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean"), 300)
    od, oi = O.knn_exact(q, y, 300, "euclidean")
    np.testing.assert_array_equal(ii.numpy(), oi)
    np.testing.assert_array_equal(dd.numpy(), od)
    assert st["n_escalated_rows"] > 0, st          # the fused pass alone could not certify these rows


def test_long_k_through_the_api_with_hubness():
    from kiez_amd import Kiez
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(3)
    s, t = rng.rand(2500, 40).astype(np.float32), rng.rand(30000, 40).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for hub in (None, "CSLS"):
            kz = Kiez(n_candidates=200, algorithm="SklearnNN", algorithm_kwargs={"metric": "euclidean"}, hubness=hub).fit(s, t)
            d, i = kz.kneighbors(150)
            od, oi = O.kiez_pipeline(s, t, 200, 150, "euclidean", 2, hub, {})
            np.testing.assert_array_equal(i, oi)
            np.testing.assert_allclose(d, od, rtol=1e-9, atol=1e-12)
