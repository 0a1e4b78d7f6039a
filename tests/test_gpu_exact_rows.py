"""The exact float64 distance kernels: the one that takes many pairs per wave step (kz_exact_dist_rows_kernel: four query rows in registers,
64 / LPR consecutive index rows per step) against the one-pair-per-wave kernel it replaces for float32 rows of d <= 512: the same
values bit for bit (both reproduce kz_wave_dot's order of operations), on every metric, ragged sizes, with and without the
normalised float64 rows of a cosine index.  The exact kernels are the backstop below every approximate tier (the reference has no
tiers: scikit-learn's brute force, sklearn_nearest_neighbors.py:96-101).  `pytest -m gpu`."""
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


@pytest.mark.parametrize("metric", ["euclidean", "sqeuclidean", "cosine"])
@pytest.mark.parametrize("d", [4, 20, 32, 64, 100, 128, 200, 256, 260, 300, 384, 512])      # (> 256: two chunks per lane, round 6)
def test_same_bits_as_the_one_pair_kernel(ctx, metric, d):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(d)
    n_q, n_i, k = (70, 3001, 7) if d != 64 else (301, 9000, 50)      # (d = 64, 301 rows: the batch that builds the float64 rows too)
    q = rng.standard_normal((n_q, d)).astype(np.float32)
    y = rng.standard_normal((n_i, d)).astype(np.float32)
    y[5] = y[17]              # exact duplicates
    y[100] = 0.0              # a zero row (cosine: norm 0 -> 1)
    qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
    ctx.set_option("eps_scale", 1e30)      # every row fails every certification: the exact kernels answer
    out = {}
    for rows in (0, 1, 2):      # one pair per wave / many pairs per wave step / one pair per LANE (kz_exact_lanes.h, round 6)
        ctx.set_option("exact_rows", rows)
        dd, ii, st = N.knn(ctx, qm, ym, k)
        assert st["n_fallback_rows"] == n_q, st
        out[rows] = (dd.numpy(), ii.numpy())
    for rows in (1, 2):
        np.testing.assert_array_equal(out[0][1], out[rows][1], err_msg=f"exact_rows {rows}")
        np.testing.assert_array_equal(out[0][0], out[rows][0], err_msg=f"exact_rows {rows}")
    q64, y64 = (q.astype(np.float64), y.astype(np.float64)) if metric == "cosine" else (q, y)
    od, oi = O.knn_exact(q64, y64, k, metric)
    np.testing.assert_array_equal(out[1][1], oi)


@pytest.mark.parametrize("d", [64, 200, 300])
def test_cosine_on_raw_rows_in_the_one_pair_per_lane_kernel(ctx, d):
    """Fewer than 64 uncertified rows: the normalised float64 image of a cosine index is not built, and the one-pair-per-lane kernel
    divides the raw rows itself (COS_RAW) -- the same bits as the other two kernels."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(100 + d)
    q = rng.standard_normal((40, d)).astype(np.float32)
    y = rng.standard_normal((5003, d)).astype(np.float32)
    y[7] = 0.0
    y[11] = y[3]
    ctx.set_option("eps_scale", 1e30)
    ctx.set_option("spec_rows", 0)         # (the ordinary exact path: 40 rows >= the lane kernel's 32, < the image's 64)
    try:
        out = {}
        for rows in (0, 1, 2):
            ctx.set_option("exact_rows", rows)
            qm, ym = N.DeviceMatrix(ctx, q, "cosine"), N.DeviceMatrix(ctx, y, "cosine")     # (fresh matrices: no image from an earlier batch)
            dd, ii, st = N.knn(ctx, qm, ym, 9)
            assert st["n_fallback_rows"] == 40, st
            out[rows] = (dd.numpy(), ii.numpy())
    finally:
        ctx.set_option("spec_rows", 64)
    for rows in (1, 2):
        np.testing.assert_array_equal(out[0][1], out[rows][1])
        np.testing.assert_array_equal(out[0][0], out[rows][0])
    np.testing.assert_array_equal(out[2][1], O.knn_exact(q.astype(np.float64), y.astype(np.float64), 9, "cosine")[1])


def test_values_are_those_of_kz_pair_values(ctx):
    """... and bit for bit the values kz_pair_values gives for the same pairs (the ordering values that travel between GPUs)."""
    from kiez_amd import _native as N
    rng = np.random.default_rng(1)
    q = rng.standard_normal((200, 96)).astype(np.float32)
    y = rng.standard_normal((5000, 96)).astype(np.float32)
    for metric in ("sqeuclidean", "cosine"):
        qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
        ctx.set_option("eps_scale", 1e30)
        dd, ii, st = N.knn(ctx, qm, ym, 9)
        ctx.set_option("eps_scale", 1.0)
        val = ctx.empty((200, 9), np.float64)
        N._check(ctx.lib.kz_pair_values(ctx.handle, qm.handle, 0, 200, ym.handle, ii.ptr, 9, val.ptr), "kz_pair_values")
        np.testing.assert_array_equal(val.numpy(), dd.numpy())       # (sqeuclidean / cosine + float64 output: the distance IS the value)


@pytest.mark.parametrize("k", [30, 64])
def test_many_exact_ties_at_the_kth_place(ctx, k):
    """Small-integer rows: hundreds of index rows at exactly the same distance -- the radix selection of the exact kernels' first level
    (k >= 24: kz_exact_chunk_radix_kernel) must keep, of the rows tied at the k-th place, those with the smallest index."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(k)
    q = rng.integers(0, 3, (150, 8)).astype(np.float32)
    y = rng.integers(0, 3, (21_000, 8)).astype(np.float32)      # (six chunks of 4 096: the two-level selection)
    qm, ym = N.DeviceMatrix(ctx, q, "sqeuclidean"), N.DeviceMatrix(ctx, y, "sqeuclidean")
    ctx.set_option("eps_scale", 1e30)
    out = {}
    for rows in (0, 1, 2):
        ctx.set_option("exact_rows", rows)
        dd, ii, st = N.knn(ctx, qm, ym, k)
        assert st["n_fallback_rows"] == 150, st
        out[rows] = (dd.numpy(), ii.numpy())
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][1], out[2][1])
    np.testing.assert_array_equal(out[0][0], out[2][0])
    np.testing.assert_array_equal(out[0][0], out[1][0])
    od, oi = O.knn_exact(q, y, k, "sqeuclidean")
    np.testing.assert_array_equal(out[1][1], oi)
    np.testing.assert_array_equal(out[1][0], od)
