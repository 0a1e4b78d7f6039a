"""BASELINE.json's full-size configurations on the GPU.  C1 / C2 (100k x 100k, d=128, k=10): EVERY row of both kNN passes and of
the final result against the oracle (2.56 Tflop of dgemm per pass: seconds on the GPU box's host cores) plus size-independent
properties; C3 (500k x 500k): 1 024-row samples of both passes, 256 rows of the MP-empiric transform; edge cases the domain has
(exact duplicates, zero rows, tiny inputs)."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c1_data():
    rng = np.random.RandomState(0)   # the reference's docstring data style (kiez/kiez.py:50-52), float32
    return rng.rand(100_000, 128).astype(np.float32), rng.rand(100_000, 128).astype(np.float32)


THREADS = max(1, min(32, os.cpu_count() or 1))   # (oracle.knn_exact works on its row chunks with this many threads)


@pytest.fixture(scope="module")
def c1_oracle_forward(c1_data):
    """The oracle's forward pass over ALL 100 000 rows -- C1's answer and the first half of C2's: computed once (22 s of host dgemm)."""
    from oracle import kiez_oracle as O
    s, t = c1_data
    return O.knn_exact(s, t, 10, "euclidean", threads=THREADS)


def _props(dist, ind, n_index, k):
    assert dist.shape == ind.shape == (dist.shape[0], k)
    assert ind.dtype == np.int64 and dist.dtype == np.float64
    assert (ind >= 0).all() and (ind < n_index).all()
    s = np.sort(ind, axis=1)
    assert (s[:, 1:] != s[:, :-1]).all(), "duplicate neighbour ids in a row"
    assert np.isfinite(dist).all()


def test_c1_full_size_every_row_against_the_oracle(c1_data, c1_oracle_forward):
    from kiez_amd import Kiez
    from oracle import kiez_oracle as O
    s, t = c1_data
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=10, algorithm="SklearnNN", algorithm_kwargs={"metric": "euclidean"}).fit(s, t)
        d10, i10 = kz.kneighbors(10)
        d5, i5 = kz.kneighbors(5)
    _props(d10, i10, len(t), 10)
    assert (np.diff(d10, axis=1) >= 0).all(), "rows must be sorted ascending"
    np.testing.assert_array_equal(i5, i10[:, :5])          # k-prefix property
    np.testing.assert_array_equal(d5, d10[:, :5])
    od, oi = c1_oracle_forward      # ALL 100 000 rows
    np.testing.assert_array_equal(i10, oi)
    np.testing.assert_array_equal(d10, od)                  # float32 inputs: bit-identical distances (sqrt rule)
    # exact distance recomputed in float64 for a few entries
    r = np.random.RandomState(1).choice(len(s), 16, replace=False)
    ref = np.sqrt(((s[r, None, :].astype(np.float64) - t[i10[r]].astype(np.float64)) ** 2).sum(-1))
    np.testing.assert_allclose(d10[r], ref, rtol=2e-7)
    assert kz.algorithm.last_stats["n_fallback_rows"] < 100


def test_c2_full_size_csls_every_row_against_the_oracle(c1_data, c1_oracle_forward):
    from kiez_amd import Kiez
    from oracle import kiez_oracle as O
    s, t = c1_data
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=10, algorithm="SklearnNN", algorithm_kwargs={"metric": "euclidean"}, hubness="CSLS").fit(s, t)
        d, i = kz.kneighbors(10)
    _props(d, i, len(t), 10)
    assert (np.diff(d, axis=1) >= 0).all()
    # the whole pipeline in the oracle, ALL rows: reverse pass (fit state), forward pass, CSLS, final sort
    rd, _ = O.knn_exact(t, s, 10, "euclidean", threads=THREADS)
    np.testing.assert_array_equal(kz.hubness._r_train_dev.numpy(), rd.mean(axis=1))       # csls.py:90, bit for bit
    fd, fi = c1_oracle_forward
    od, oi = O.sort_topk(O.csls_transform(fd, fi, rd), fi, 10)
    np.testing.assert_array_equal(i, oi)
    np.testing.assert_allclose(d, od, rtol=1e-9, atol=1e-12)


def test_exact_duplicates_order_by_index():
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(4)
    base = rng.rand(50, 16)
    t = np.vstack([base, base, base[:10]])          # exact copies -> exact distance ties
    s = rng.rand(40, 16)
    ctx = N.Context.get()
    d, i, st = N.knn(ctx, N.DeviceMatrix(ctx, s, "euclidean"), N.DeviceMatrix(ctx, t, "euclidean"), 12)
    od, oi = O.knn_exact(s, t, 12, "euclidean")
    np.testing.assert_array_equal(i.numpy(), oi)
    np.testing.assert_allclose(d.numpy(), od, rtol=1e-12, atol=1e-12)


def test_cosine_with_zero_rows_and_tiny_inputs():
    from kiez_amd import Kiez
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(8)
    s, t = rng.rand(7, 3), rng.rand(9, 3)
    t[4] = 0.0                                        # sklearn normalize(): a zero row stays zero -> distance 1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for hub in (None, "CSLS", "LocalScaling"):
            d, i = Kiez(n_candidates=4, algorithm="SklearnNN", algorithm_kwargs={"metric": "cosine"}, hubness=hub).fit(s, t).kneighbors(3)
            od, oi = O.kiez_pipeline(s, t, 4, 3, "cosine", 2, hub, {})
            np.testing.assert_array_equal(i, oi)
            np.testing.assert_allclose(d, od, rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):   # empty matrices are rejected (sklearn: "Found array with 0 sample(s)")
        Kiez(hubness="CSLS").fit(np.zeros((0, 3)), t)
    with pytest.raises(ValueError):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Kiez().fit(np.zeros((0, 3)), t).kneighbors(1)


def test_float32_cosine_keeps_reference_output_dtype():
    from kiez_amd import Kiez
    rng = np.random.RandomState(2)
    s, t = rng.rand(50, 8).astype(np.float32), rng.rand(60, 8).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d, i = Kiez(n_candidates=5, algorithm="SklearnNN", algorithm_kwargs={"metric": "cosine"}).fit(s, t).kneighbors(5)
        d2, _ = Kiez(n_candidates=5, algorithm="SklearnNN", algorithm_kwargs={"metric": "cosine"}, hubness="CSLS").fit(s, t).kneighbors(5)
        d3, _ = Kiez(n_candidates=5, algorithm="SklearnNN", algorithm_kwargs={"metric": "euclidean"}).fit(s, t).kneighbors(5)
    assert d.dtype == np.float32 and d2.dtype == np.float32 and d3.dtype == np.float64 and i.dtype == np.int64


def test_long_index_keeps_the_dma_ring_honest():
    """1M index rows x 300 features (C4's index): 7813 index tiles per sweep through the one-workgroup-per-CU kernel, whose
    LDS-DMA ring waits with a COUNTED s_waitcnt.  A count that forgot the one-slice fragment prefetch let a wave read a slot
    before its DMA had landed: 3 wrong rows in 3000 -- only at this length, and visible in kz_knn_stats.max_err_ratio."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(0)
    t = rng.random((1_000_000, 300), dtype=np.float32)
    s = rng.random((2048, 300), dtype=np.float32)
    ctx = N.Context.get()
    ym = N.DeviceMatrix(ctx, t, "euclidean")
    qm = N.DeviceMatrix(ctx, s, "euclidean")
    worst = 0.0
    for _ in range(3):   # the race was timing dependent
        d, i, st = N.knn(ctx, qm, ym, 10)
        worst = max(worst, st["max_err_ratio"])
    assert worst < 0.5, st
    od, oi = O.knn_exact(s[:256], t, 10, "euclidean")
    np.testing.assert_array_equal(i.numpy()[:256], oi)
    np.testing.assert_array_equal(d.numpy()[:256], od)


@pytest.mark.parametrize("n_s,n_t,d,metric,k", [
    (100_000, 100_000, 128, "euclidean", 10),     # C1: fp16 kernel at three workgroups per CU, lists in LDS
    (30_000, 300_000, 300, "euclidean", 10),      # C4-shaped: two workgroups per CU, 19 slices
    (40_000, 200_000, 200, "cosine", 50),         # C3-shaped: list length 64 (lists in the output arrays), 13 slices
])
def test_two_independent_kernels_agree_on_every_row(n_s, n_t, d, metric, k):
    """The fp16 kernel (default first pass) and the float32-operand kernel share no synchronisation structure (LDS-DMA ring
    + event pool vs register staging + one barrier per slice).  At full length they must agree on EVERY row, bit for bit; a
    race in either shows up here even when it only drops a true neighbour (which the bound self-check cannot see)."""
    from kiez_amd import _native as N
    rng = np.random.default_rng(d)
    t = rng.random((n_t, d), dtype=np.float32)
    s = rng.random((n_s, d), dtype=np.float32)
    if metric == "cosine":
        t, s = t.astype(np.float64), s.astype(np.float64)
    ctx = N.Context.get()
    res = {}
    for prec in (0, 1):
        ctx.set_option("precision", prec)
        try:
            ym, qm = N.DeviceMatrix(ctx, t, metric), N.DeviceMatrix(ctx, s, metric)
            for _ in range(2):   # races are timing dependent: two passes each
                dd, ii, st = N.knn(ctx, qm, ym, k)
                assert st["max_err_ratio"] < 0.5, st
                if prec in res:
                    np.testing.assert_array_equal(res[prec][1], ii.numpy())
                res[prec] = (dd.numpy(), ii.numpy())
        finally:
            ctx.set_option("precision", 0)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][0], res[1][0])


def test_c3_full_size_mp_empiric_properties_and_oracle_sample():
    """BASELINE config 3 at FULL size through the drop-in API: 500k x 500k, d=200, cosine, k=50, MutualProximity empiric.
    The oracle cannot run this size (the reference's own loop is ~2.5e7 allocations of 4 MB), so:
      * size-independent properties of the full result;
      * a 1 024-row sample of BOTH kNN passes against the oracle's exact float64 search;
      * the oracle's MP-empiric transform + final sort on 256 rows, fed with the (sample-verified) reverse lists."""
    from kiez_amd import Kiez
    from oracle import kiez_oracle as O
    from tests.golden_util import knife_edge_rows, knife_edge_topk_ok
    n, d, K = 500_000, 200, 50
    rng = np.random.RandomState(0)
    s = rng.rand(n, d).astype(np.float32)
    t = rng.rand(n, d).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=K, algorithm="SklearnNN", algorithm_kwargs={"metric": "cosine"}, hubness="MutualProximity",
                  hubness_kwargs={"method": "empiric"})
        kz.fit(s, t)
        dist, ind = kz.kneighbors(K)
    # properties
    assert dist.shape == (n, K) and ind.shape == (n, K) and dist.dtype == np.float32 and ind.dtype == np.int64
    assert ind.min() >= 0 and ind.max() < n
    assert (np.diff(dist, axis=1) >= 0).all()                       # ascending
    assert dist.min() >= 0.0 and dist.max() <= 1.0                  # 1 - count / K
    assert np.allclose(dist * K, np.round(dist * K), atol=1e-4)     # multiples of 1/K
    srt = np.sort(ind, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                        # distinct neighbours per row
    # both passes on a row sample (float64 casts: the oracle's convention for cosine + float32)
    nn = kz.algorithm
    rows = np.arange(0, n, n // 1024)[:1024]
    s64, t64 = s.astype(np.float64), None
    fd, fi = nn.kneighbors_device(k=K)
    fd, fi = fd.numpy(), fi.numpy()
    rd_dev, ri_dev = kz.hubness._dist_t2s_dev.numpy(), kz.hubness._ind_t2s_dev.numpy()
    t64 = t.astype(np.float64)
    od, oi = O.knn_exact(s64[rows], t64, K, "cosine", threads=THREADS)
    np.testing.assert_array_equal(fi[rows], oi)
    np.testing.assert_allclose(fd[rows], od, rtol=1e-6, atol=1e-7)
    od, oi = O.knn_exact(t64[rows], s64, K, "cosine", threads=THREADS)
    np.testing.assert_array_equal(ri_dev[rows], oi)
    np.testing.assert_allclose(rd_dev[rows], od, rtol=1e-6, atol=1e-7)
    # transform + sort on 256 of those rows (the oracle's per-row loop)
    sub = rows[::4]
    hr = O.mp_empiric_transform(fd[sub], fi[sub], rd_dev, ri_dev)
    sd, si = O.sort_topk(hr, fi[sub], K)
    ke = (fi[sub] == sub[:, None]).any(axis=1)
    for a, r in enumerate(sub):
        if ke[a]:
            assert knife_edge_topk_ok(sd[a], si[a], dist[r].astype(np.float64), ind[r], r, K, ri_dev), r
        else:
            np.testing.assert_array_equal(ind[r], si[a])
            np.testing.assert_allclose(dist[r], sd[a], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("d", [64, 200, 72])
def test_long_sweeps_of_the_fp16_kernel_at_three_workgroups_per_cu(d):
    """1563 index tiles per sweep through the three-workgroups-per-CU builds (4-slot DMA ring with hand-issued saddr-form
    LDS-DMA; d = 200: single fragment set): must agree with the float32-operand kernel (a different synchronisation
    structure) on every row and with the oracle -- even, odd and 13 slice counts."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(d)
    t = rng.random((200_000, d), dtype=np.float32)
    s = rng.random((6000, d), dtype=np.float32)
    ctx = N.Context.get()
    ym, qm = N.DeviceMatrix(ctx, t, "euclidean"), N.DeviceMatrix(ctx, s, "euclidean")
    res = {}
    try:
        for name, prec in (("fp16", 0), ("f32", 1)):
            ctx.set_option("precision", prec)
            for _ in range(2):   # races are timing dependent
                dd, ii, st = N.knn(ctx, qm, ym, 10)
                assert st["max_err_ratio"] < 0.6, (name, st)
            res[name] = (dd.numpy(), ii.numpy())
    finally:
        ctx.set_option("precision", 0)
    np.testing.assert_array_equal(res["fp16"][1], res["f32"][1])
    np.testing.assert_array_equal(res["fp16"][0], res["f32"][0])
    od, oi = O.knn_exact(s[:256], t, 10, "euclidean")
    np.testing.assert_array_equal(res["fp16"][1][:256], oi)
