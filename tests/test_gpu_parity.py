"""Parity of the HIP path (through the C ABI) with the golden vectors of the real reference and with the
oracle on seeded inputs.  Needs an MI355X: `pytest -m gpu`."""
import warnings

import numpy as np
import pytest

from tests.golden_util import (GOLDEN, HUB, case_params, knife_edge_rows, knife_edge_topk_ok, knife_edge_transform_ok, ktag,
                               load_case)

pytestmark = pytest.mark.gpu

# A/B runs of the suite pin a precision tier through the environment (KZ_PRECISION=fp32|bf16); the tests that assert WHICH
# tier ran only make sense for the default selection.
import os
_PINNED = bool(os.environ.get("KZ_PRECISION"))
TIER_F32, TIER_BF16, TIER_FP16 = 0, 1, 2          # kz_knn_stats.first_pass
PREC_FP16, PREC_F32, PREC_BF16 = 0, 1, 2          # context option "precision"
default_tiers_only = pytest.mark.skipif(_PINNED, reason="a kernel is pinned through the environment")

RTOL = 1e-5   # north-star tolerance for rescaled distances
ATOL = 1e-6   # self distances of a single-source reverse pass: exact 0 here vs sqrt(1e-14) in sklearn


def _kiez(K, metric, p, hname, kw, **algo_kw):
    from kiez_amd import Kiez
    return Kiez(n_candidates=K, algorithm="SklearnNN", algorithm_kwargs=dict(metric=metric, p=p, **algo_kw),
                hubness=hname, hubness_kwargs=dict(kw))


_knife_edge_rows = knife_edge_rows   # (rows compared tie-tolerantly, never dropped: tests/golden_util.py)


@pytest.mark.parametrize("case,tag,k", case_params())
def test_golden_pipeline(case, tag, k):
    g = load_case(case)
    hname, kw = HUB[tag]
    src, tgt = g["source"], g["_target"]
    if g["_metric"] == "cosine":
        pass  # goldens were generated from float64 casts of float32 data; we feed the same float64 arrays
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = _kiez(g["_K"], g["_metric"], g["_p"], hname, kw)
        kz.fit(src, tgt)
        d, i = kz.kneighbors(k)
    ref_d, ref_i = g[f"{tag}__k{ktag(k)}__dist"], g[f"{tag}__k{ktag(k)}__ind"]
    assert i.dtype == np.int64 and i.shape == ref_i.shape and d.shape == ref_d.shape
    keep = np.ones(len(i), dtype=bool)
    if tag == "mp_empiric":
        keep &= ~_knife_edge_rows(g["mp_empiric__ind_s2t"])
        for r in np.flatnonzero(~keep):
            assert knife_edge_topk_ok(ref_d[r], ref_i[r], d[r], i[r], r, g["_K"], g["mp_empiric__ind_t2s"]), f"knife-edge row {r}"
    np.testing.assert_array_equal(i[keep], ref_i[keep])
    rtol, atol = RTOL, ATOL   # (DisSimLocal included: float64 on the device against the reference's float32-mixed arithmetic
                              #  stays within the north-star 1e-5 relative; ATOL is the single-source self-distance floor)
    np.testing.assert_allclose(d[keep], ref_d[keep], rtol=rtol, atol=atol)


@pytest.mark.parametrize("case", ["c0_two_source", "f32_euclidean", "cosine_k50", "conftest_single_source"])
def test_golden_intermediates(case):
    """Stage-by-stage: reverse kNN, forward kNN and the unsorted transform output."""
    from kiez_amd import hubness_reduction as H
    g = load_case(case)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag in g["_tags"]:
            if tag == "none":
                continue
            hname, kw = HUB[tag]
            kz = _kiez(g["_K"], g["_metric"], g["_p"], hname, kw)
            kz.fit(g["source"], g["_target"])
            nn = kz.algorithm
            tgt = g["source"] if g["_target"] is None else g["_target"]
            d_t2s, i_t2s = nn.kneighbors(k=g["_K"], query=tgt, s_to_t=False)
            d_s2t, i_s2t = nn.kneighbors(k=g["_K"])
            np.testing.assert_array_equal(i_t2s, g[f"{tag}__ind_t2s"])
            np.testing.assert_array_equal(i_s2t, g[f"{tag}__ind_s2t"])
            np.testing.assert_allclose(d_t2s, g[f"{tag}__dist_t2s"], rtol=1e-9, atol=ATOL)
            np.testing.assert_allclose(d_s2t, g[f"{tag}__dist_s2t"], rtol=1e-9, atol=ATOL)
            tr, _ = kz.hubness.transform(g[f"{tag}__dist_s2t"], g[f"{tag}__ind_s2t"], g["source"])
            keep = np.ones(len(tr), dtype=bool)
            if tag == "mp_empiric":
                keep &= ~_knife_edge_rows(g["mp_empiric__ind_s2t"])
                for r in np.flatnonzero(~keep):
                    assert knife_edge_transform_ok(g[f"{tag}__transformed"][r], tr[r], g[f"{tag}__ind_s2t"][r], r, g["_K"],
                                                   g[f"{tag}__ind_t2s"]), r
            np.testing.assert_allclose(tr[keep], g[f"{tag}__transformed"][keep], rtol=RTOL,
                                       atol=ATOL)


@pytest.mark.parametrize("k", [1, 2, 3, 5, 10])
def test_golden_sort(k):
    from kiez_amd.hubness_reduction import HubnessReduction
    z = np.load(GOLDEN / "sort.npz")
    d, i = HubnessReduction._sort(z["dist0"], z["ind0"], k)
    np.testing.assert_array_equal(i, z[f"sorted0_k{k}_ind"])
    np.testing.assert_array_equal(d, z[f"sorted0_k{k}_dist"])
    d, i = HubnessReduction._sort(z["tie_d"], z["tie_i"], k)
    np.testing.assert_array_equal(d, z[f"tie_k{k}_dist"])
    np.testing.assert_array_equal(i, z[f"tie_k{k}_ind"])


# ---------------------------------------------------------------------------------------------------
# seeded inputs against the oracle
# ---------------------------------------------------------------------------------------------------
def _data(n_s, n_t, d, dtype, seed=0, gauss=False):
    rng = np.random.RandomState(seed)
    gen = rng.randn if gauss else rng.rand
    return gen(n_s, d).astype(dtype), gen(n_t, d).astype(dtype)


@pytest.mark.parametrize("n_s,n_t,d,dtype,metric,K", [
    (1000, 1300, 64, np.float32, "euclidean", 10),
    (777, 513, 50, np.float64, "minkowski", 10),
    (2000, 3000, 200, np.float32, "cosine", 50),
    (1500, 1500, 128, np.float32, "sqeuclidean", 5),
    (300, 5000, 300, np.float32, "euclidean", 10),
    (4100, 130, 17, np.float64, "euclidean", 25),
])
@pytest.mark.parametrize("tag", ["none", "csls", "ls", "nicdm", "mp_normal", "mp_empiric", "dsl"])
def test_oracle_parity(n_s, n_t, d, dtype, metric, K, tag):
    from oracle import kiez_oracle as O
    if tag == "dsl" and metric == "cosine":
        pytest.skip("DSL rejects cosine (dis_sim.py:47-61)")
    hname, kw = HUB[tag]
    s, t = _data(n_s, n_t, d, dtype, seed=n_s + d)
    if metric == "cosine":
        s, t = s.astype(np.float64), t.astype(np.float64)
    k = max(1, K // 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = _kiez(K, metric, 2, hname, kw)
        kz.fit(s, t)
        dist, ind = kz.kneighbors(k)
    # the oracle's MP-empiric transform is a Python double loop: at K = 50 it is evaluated on a row sample (the fit state
    # still comes from ALL rows: the transform is row-local given the fit state)
    rows = 300 if (tag == "mp_empiric" and n_s * K * K > 4e6) else None
    od, oi = O.kiez_pipeline(s, t, K, k, metric, 2, hname, kw, query_rows=rows)
    dist, ind = dist[:len(oi)], ind[:len(oi)]
    keep = np.ones(len(ind), dtype=bool)
    if tag == "mp_empiric":
        keep &= ~_knife_edge_rows(O.knn_exact(s[:len(oi)], t, K, O.canonical_metric(metric))[1])
        if (~keep).any():
            ind_t2s = O.knn_exact(t, s, min(K, len(s)), O.canonical_metric(metric))[1]
            for r in np.flatnonzero(~keep):
                assert knife_edge_topk_ok(od[r], oi[r], dist[r], ind[r], r, K, ind_t2s), f"knife-edge row {r}"
    bad = (ind != oi).any(axis=1) & keep
    assert not bad.any(), f"{bad.sum()} rows differ"
    np.testing.assert_allclose(dist[keep], od[keep], rtol=RTOL, atol=ATOL)


def test_single_source_self_is_stripped_forward_but_kept_reverse():
    from oracle import kiez_oracle as O
    s, _ = _data(1500, 1, 40, np.float32, seed=5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = _kiez(10, "euclidean", 2, "CSLS", {})
        kz.fit(s)
        d, i = kz.kneighbors(10)
        fwd_d, fwd_i = kz.algorithm.kneighbors(k=10)
        rev_d, rev_i = kz.algorithm.kneighbors(k=10, query=s, s_to_t=False)
    assert not (fwd_i == np.arange(len(s))[:, None]).any()
    assert (rev_i[:, 0] == np.arange(len(s))).all() and (rev_d[:, 0] == 0).all()
    od, oi = O.kiez_pipeline(s, None, 10, 10, "euclidean", 2, "CSLS", {})
    np.testing.assert_array_equal(i, oi)
    np.testing.assert_allclose(d, od, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("splits", [1, 2, 5])
def test_split_counts_give_identical_results(splits):
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    s, t = _data(700, 5000, 48, np.float32, seed=9)
    ctx = N.Context.get()
    ctx.set_option("force_splits", splits)
    try:
        qm, ym = N.DeviceMatrix(ctx, s, "euclidean"), N.DeviceMatrix(ctx, t, "euclidean")
        d, i, st = N.knn(ctx, qm, ym, 10)
        assert st["n_splits"] == splits
        od, oi = O.knn_exact(s, t, 10, "euclidean")
        np.testing.assert_array_equal(i.numpy(), oi)
        np.testing.assert_allclose(d.numpy(), od, rtol=1e-9, atol=1e-9)
    finally:
        ctx.set_option("force_splits", 0)


def test_exact_fallback_path_is_exact():
    """eps_scale = 1e9 makes every row fail certification: the float64 brute-force backstop must agree too."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    ctx = N.Context.get()
    for metric, dtype in (("euclidean", np.float32), ("cosine", np.float64), ("sqeuclidean", np.float64)):
        s, t = _data(150, 900, 33, dtype, seed=21)
        ctx.set_option("eps_scale", 1e9)
        try:
            qm, ym = N.DeviceMatrix(ctx, s, metric), N.DeviceMatrix(ctx, t, metric)
            d, i, st = N.knn(ctx, qm, ym, 7)
            assert st["n_fallback_rows"] == 150
        finally:
            ctx.set_option("eps_scale", 1.0)
        od, oi = O.knn_exact(s, t, 7, metric)
        np.testing.assert_array_equal(i.numpy(), oi)
        np.testing.assert_allclose(d.numpy(), od, rtol=1e-9, atol=1e-9)
        # self-stripping through the fallback
        ctx.set_option("eps_scale", 1e9)
        try:
            d, i, st = N.knn(ctx, ym, ym, 7, exclude_self=True)
        finally:
            ctx.set_option("eps_scale", 1.0)
        od, oi = O.knn_exact(t, t, 7, metric, exclude_self=True)
        np.testing.assert_array_equal(i.numpy(), oi)


def test_near_duplicates_trigger_certification_failure_but_stay_exact():
    """Index rows that differ by ~1e-7 relative cannot be separated in float32: those queries must go down the tiers
    and still return the float64 order."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(3)
    base = rng.rand(40, 32)
    t = np.repeat(base, 30, axis=0) + 1e-9 * rng.rand(1200, 32)   # 30 near-copies of each of 40 points, float64
    s = rng.rand(64, 32)
    ctx = N.Context.get()
    qm, ym = N.DeviceMatrix(ctx, s, "euclidean"), N.DeviceMatrix(ctx, t, "euclidean")
    d, i, st = N.knn(ctx, qm, ym, 10)
    # (lists of 16 cannot hold the 30 copies of the nearest point: every query goes down -- as far as lists of 64, which hold the
    #  copies of the two nearest points and are certified against the third; the float64 re-rank orders the copies)
    assert st["n_escalated_rows"] + st["n_fallback_rows"] >= 64
    od, oi = O.knn_exact(s, t, 10, "euclidean")
    # float64 expanded-form distances of near-copies agree to ~1e-16 relative: compare as sets per tie group
    dd = d.numpy()
    np.testing.assert_allclose(dd, od, rtol=1e-7, atol=1e-9)
    assert (np.sort(i.numpy() // 30, axis=1) == np.sort(oi // 30, axis=1)).all()


def test_medium_size_all_rows_bit_identical():
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    s, t = _data(20000, 20000, 128, np.float32, seed=0)
    ctx = N.Context.get()
    qm, ym = N.DeviceMatrix(ctx, s, "euclidean"), N.DeviceMatrix(ctx, t, "euclidean")
    d, i, st = N.knn(ctx, qm, ym, 10)
    od, oi = O.knn_exact(s, t, 10, "euclidean")
    assert np.array_equal(i.numpy(), oi)
    np.testing.assert_allclose(d.numpy(), od, rtol=1e-9, atol=0)
    assert st["n_fallback_rows"] < 20


def test_query_chunking_and_row_ranges():
    """kz_knn processes long query sets in chunks and accepts a row range (multi-GPU sharding): both must not change results."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    s, t = _data(1000, 700, 24, np.float32, seed=33)
    ctx = N.Context.get()
    ym = N.DeviceMatrix(ctx, t, "euclidean")
    qm = N.DeviceMatrix(ctx, s, "euclidean")
    od, oi = O.knn_exact(s, t, 10, "euclidean")
    ctx.set_option("chunk_rows", 300)   # not a multiple of the 128-row tile
    try:
        d, i, _ = N.knn(ctx, qm, ym, 10)
        np.testing.assert_array_equal(i.numpy(), oi)
        d2, i2, _ = N.knn(ctx, qm, ym, 10, q_begin=137, q_count=555)
        np.testing.assert_array_equal(i2.numpy(), oi[137:137 + 555])
        np.testing.assert_array_equal(d2.numpy(), od[137:137 + 555])
        # self-stripping with a row range (global row ids)
        dm, im, _ = N.knn(ctx, ym, ym, 5, exclude_self=True, q_begin=129, q_count=400)
        sd, si = O.knn_exact(t, t, 5, "euclidean", exclude_self=True)
        np.testing.assert_array_equal(im.numpy(), si[129:529])
    finally:
        ctx.set_option("chunk_rows", 0)


@default_tiers_only
@pytest.mark.parametrize("n_s,n_t,d,dtype,metric,k,single", [
    (900, 2100, 128, np.float32, "euclidean", 10, False),   # 8 slices: pipelined pairs
    (700, 1500, 40, np.float64, "sqeuclidean", 7, False),   # 3 slices: odd count, barrier parity changes per tile
    (1300, 1300, 96, np.float32, "cosine", 25, True),       # list length 32, self stripped
    (400, 5000, 17, np.float32, "euclidean", 50, False),    # 2 slices, list length 64
    (500, 1900, 70, np.float32, "euclidean", 10, False),    # 5 slices
    (500, 1900, 100, np.float64, "euclidean", 100, False),  # 7 slices, list length 128
    (640, 2500, 384, np.float32, "sqeuclidean", 10, False), # 24 slices: the largest stationary query tile
    (300, 800, 16, np.float32, "euclidean", 5, False),      # 1 slice: 16-bit tiers not eligible, float32 kernel
    (300, 800, 200, np.float32, "euclidean", 5, False),     # 13 slices: one workgroup per CU, odd slice count
    (400, 1700, 224, np.float32, "euclidean", 10, False),   # 14 slices: two workgroups per CU, single fragment set
    (400, 1700, 256, np.float64, "sqeuclidean", 27, True),  # 16 slices: the last shape on the two-workgroup kernel
    (260, 900, 300, np.float64, "cosine", 10, False),       # 19 slices
    (200, 600, 400, np.float32, "euclidean", 5, False),     # 26 slices: not eligible, float32 kernel
])
def test_precision_tiers_agree_bit_for_bit(n_s, n_t, d, dtype, metric, k, single):
    """The fp16 first pass (default), the split-bf16 pass and the float32-operand kernel must return the same float64 answer."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    s, t = _data(n_s, n_t, d, dtype, seed=d + k)
    if single:
        t = s
    ctx = N.Context.get()
    res = {}
    for prec in (PREC_FP16, PREC_BF16, PREC_F32):
        ctx.set_option("precision", prec)
        try:
            qm = N.DeviceMatrix(ctx, s, metric)
            ym = qm if single else N.DeviceMatrix(ctx, t, metric)
            dd, ii, st = N.knn(ctx, qm, ym, k, exclude_self=single)
            res[prec] = (dd.numpy(), ii.numpy(), st)
        finally:
            ctx.set_option("precision", 0)
    n_slices = (d + 15) // 16
    eligible = 2 <= n_slices <= 24
    assert res[PREC_FP16][2]["first_pass"] == (TIER_FP16 if eligible else TIER_F32)
    assert res[PREC_BF16][2]["first_pass"] == (TIER_BF16 if eligible else TIER_F32)
    assert res[PREC_F32][2]["first_pass"] == TIER_F32
    for prec in (PREC_BF16, PREC_F32):
        np.testing.assert_array_equal(res[PREC_FP16][1], res[prec][1])
        np.testing.assert_array_equal(res[PREC_FP16][0], res[prec][0])
    od, oi = O.knn_exact(s, t, k, O.canonical_metric(metric), exclude_self=single)
    np.testing.assert_array_equal(res[PREC_FP16][1], oi)


@default_tiers_only
def test_bf16_pass_escalates_to_float32_operands_on_tight_clusters():
    """A tight cluster far from the origin: neighbour gaps sit between the float32 and the split-bf16 rounding bounds,
    so most rows fail the bf16 certification; the chunk must be re-done with float32 operands and stay exact."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(17)
    centre = rng.rand(32)
    centre *= 10.0 / np.linalg.norm(centre)
    t = (centre + 0.05 * rng.randn(3000, 32)).astype(np.float32)
    s = (centre + 0.05 * rng.randn(500, 32)).astype(np.float32)
    ctx = N.Context.get()
    qm, ym = N.DeviceMatrix(ctx, s, "euclidean"), N.DeviceMatrix(ctx, t, "euclidean")
    ctx.set_option("precision", PREC_BF16)
    try:
        d, i, st = N.knn(ctx, qm, ym, 10)
    finally:
        ctx.set_option("precision", 0)
    assert st["n_escalated_rows"] == 500 and st["first_pass"] == TIER_BF16
    od, oi = O.knn_exact(s, t, 10, "euclidean")
    np.testing.assert_array_equal(i.numpy(), oi)
    np.testing.assert_allclose(d.numpy(), od, rtol=1e-9, atol=0)
    # the float32-only setting gives the same answer without the detour
    ctx.set_option("precision", PREC_F32)
    try:
        d1, i1, st1 = N.knn(ctx, qm, ym, 10)
    finally:
        ctx.set_option("precision", 0)
    assert st1["n_escalated_rows"] == 0
    np.testing.assert_array_equal(i1.numpy(), i.numpy())
    np.testing.assert_array_equal(d1.numpy(), d.numpy())
    # the default fp16 pass works on CENTRED operands: the offset that defeats split-bf16 is gone, nothing escalates
    d2, i2, st2 = N.knn(ctx, qm, ym, 10)
    assert st2["first_pass"] == TIER_FP16 and st2["n_escalated_rows"] == 0 and st2["n_fallback_rows"] == st2["n_spec_rows"]
    np.testing.assert_array_equal(i2.numpy(), i.numpy())
    np.testing.assert_array_equal(d2.numpy(), d.numpy())


@default_tiers_only
@pytest.mark.parametrize("single", [False, True])
def test_bf16_pass_escalates_only_the_uncertified_rows(single):
    """Uniform data plus one tight far-away cluster: only the cluster's queries fail the split-bf16 certification; they
    alone are gathered and re-done with float32 operands (self-stripping by explicit row id in single-source mode)."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(23)
    centre = rng.rand(48)
    centre *= 12.0 / np.linalg.norm(centre)
    t = np.vstack([rng.rand(3000, 48), centre + 0.05 * rng.randn(300, 48)]).astype(np.float32)
    perm = rng.permutation(len(t))
    t = t[perm]
    if single:
        s = t
    else:
        s = np.vstack([rng.rand(900, 48), centre + 0.05 * rng.randn(100, 48)]).astype(np.float32)
    ctx = N.Context.get()
    ym = N.DeviceMatrix(ctx, t, "euclidean")
    qm = ym if single else N.DeviceMatrix(ctx, s, "euclidean")
    ctx.set_option("precision", PREC_BF16)
    try:
        d, i, st = N.knn(ctx, qm, ym, 10, exclude_self=single)
    finally:
        ctx.set_option("precision", 0)
    assert st["first_pass"] == TIER_BF16
    assert 0 < st["n_escalated_rows"] <= (300 if single else 100)
    od, oi = O.knn_exact(s, t, 10, "euclidean", exclude_self=single)
    np.testing.assert_array_equal(i.numpy(), oi)
    np.testing.assert_allclose(d.numpy(), od, rtol=1e-9, atol=0)


@default_tiers_only
@pytest.mark.parametrize("single", [False, True])
def test_fp16_pass_sends_only_the_uncertified_rows_down(single):
    """Uniform data plus one VERY tight cluster (neighbour gaps far below the fp16 operand rounding): only the cluster's
    queries fail the fp16 certification; they alone go down the tiers (split-bf16 / float32 operands / exact float64) and the result
    is still the float64 order."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(29)
    centre = rng.rand(48)
    t = np.vstack([rng.rand(3000, 48), centre + 2e-3 * rng.randn(300, 48)]).astype(np.float32)
    t = t[rng.permutation(len(t))]
    if single:
        s = t
    else:
        s = np.vstack([rng.rand(900, 48), centre + 2e-3 * rng.randn(100, 48)]).astype(np.float32)
    ctx = N.Context.get()
    ym = N.DeviceMatrix(ctx, t, "euclidean")
    qm = ym if single else N.DeviceMatrix(ctx, s, "euclidean")
    d, i, st = N.knn(ctx, qm, ym, 10, exclude_self=single)
    assert st["first_pass"] == TIER_FP16
    # (nested count: more lists of 16 -> lists of 64 -> of 128 -> split-bf16 operands -> float32 operands -> exact float64, the
    #  cluster's queries at every level they reach -- and nobody else's)
    assert 0 < st["n_escalated_rows"] + st["n_fallback_rows"] <= 6 * (300 if single else 100) + 60
    od, oi = O.knn_exact(s, t, 10, "euclidean", exclude_self=single)
    np.testing.assert_array_equal(i.numpy(), oi)
    np.testing.assert_allclose(d.numpy(), od, rtol=1e-9, atol=0)


@pytest.mark.parametrize("precision", [PREC_FP16, PREC_BF16, PREC_F32])
def test_data_below_the_float32_product_range_stays_exact(precision):
    """Rows at the 1e-20 scale: q.y underflows in float32, so no unscaled approximate key can be trusted: under the
    float32 / split-bf16 tiers every row must take the exact float64 path; the fp16 image is scaled into range by a power
    of two and certifies them directly.  Same answer either way."""
    from kiez_amd import _native as N
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(5)
    s, t = 1e-20 * rng.rand(200, 24), 1e-20 * rng.rand(900, 24)
    ctx = N.Context.get()
    ctx.set_option("precision", precision)
    try:
        qm, ym = N.DeviceMatrix(ctx, s, "sqeuclidean"), N.DeviceMatrix(ctx, t, "sqeuclidean")
        d, i, st = N.knn(ctx, qm, ym, 5)
    finally:
        ctx.set_option("precision", 0)
    if precision != PREC_FP16 and not _PINNED:
        assert st["n_fallback_rows"] == 200
    od, oi = O.knn_exact(s, t, 5, "sqeuclidean")
    np.testing.assert_array_equal(i.numpy(), oi)
    np.testing.assert_allclose(d.numpy(), od, rtol=1e-9, atol=0)


@default_tiers_only
@pytest.mark.parametrize("n_s,n_t,d,dtype,metric,gauss", [
    (2000, 6000, 128, np.float32, "euclidean", False),
    (2000, 6000, 128, np.float32, "euclidean", True),
    (1500, 4000, 300, np.float64, "cosine", True),
    (1500, 4000, 64, np.float64, "sqeuclidean", False),
])
@pytest.mark.parametrize("precision", [PREC_FP16, PREC_BF16, PREC_F32])
def test_rounding_bound_holds_with_margin(n_s, n_t, d, dtype, metric, gauss, precision):
    """The certification rests on |approximate key - exact key| <= eps.  kz_knn measures the worst ratio over every
    re-ranked candidate (tens of thousands of keys per call): it must stay well below 1 for every operand precision
    (the fp16 bound is built from MEASURED residual norms, so it sits closer to the observed error than the a-priori
    float32 / split-bf16 bounds do: 0.2-0.3 against 0.005-0.03)."""
    from kiez_amd import _native as N
    s, t = _data(n_s, n_t, d, dtype, seed=3 * d, gauss=gauss)
    ctx = N.Context.get()
    ctx.set_option("precision", precision)
    try:
        qm, ym = N.DeviceMatrix(ctx, s, metric), N.DeviceMatrix(ctx, t, metric)
        _, _, st = N.knn(ctx, qm, ym, 10)
    finally:
        ctx.set_option("precision", 0)
    assert 0.0 < st["max_err_ratio"] < (0.6 if precision == PREC_FP16 else 0.5), st


def test_sort_keeps_rows_with_nan_a_permutation():
    """ADVICE (round 1): a transformed row can hold NaN (MP-normal with sd = 0, LS / NICDM with radius 0).  numpy's
    argpartition sorts NaN last and keeps the row a permutation; so must the device sort (NaN last, finite part exactly the
    selection sort of the finite values in their original order)."""
    from kiez_amd.hubness_reduction import HubnessReduction
    from oracle import kiez_oracle as O
    rng = np.random.RandomState(4)
    d = rng.rand(300, 12)
    d[rng.rand(300, 12) < 0.15] = np.nan
    d[7, :] = np.nan
    d[::5] = np.round(d[::5], 1)        # ties among the finite values
    ind = np.tile(np.arange(12, dtype=np.int64), (300, 1)) + 100
    for k in (1, 5, 12):
        od, oi = HubnessReduction._sort(d, ind, k)
        for r in range(300):
            fin = ~np.isnan(d[r])
            nf = int(fin.sum())
            assert len(set(oi[r].tolist())) == k                              # no id duplicated or dropped
            assert not np.isnan(od[r][:min(k, nf)]).any() and np.isnan(od[r][min(k, nf):]).all()   # NaN last
            if nf:
                ed, ei = O.sort_topk(d[r][fin][None, :], ind[r][fin][None, :], min(k, nf))
                # NaN never takes part in a swap before the finite values are exhausted only if it is never selected;
                # values agree with the finite-only sort (ids may differ where a NaN was swapped through: compare values)
                np.testing.assert_array_equal(od[r][:min(k, nf)], ed[0])


@pytest.mark.parametrize("metric,single", [("euclidean", False), ("cosine", True)])
def test_more_than_110_neighbours_run_on_the_exact_route(metric, single):
    """The reference's SklearnNN accepts any k <= n (sklearn_nearest_neighbors.py:51-65).  The fused kernels keep at most
    110 candidates per query; beyond that kz_knn runs every row on the exact float64 kernels -- slow, and still the
    reference's order."""
    from kiez_amd import Kiez
    from oracle import kiez_oracle as O
    s, t = _data(400, 900, 20, np.float64, seed=77)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=150, algorithm="SklearnNN", algorithm_kwargs={"metric": metric})
        kz.fit(s, None if single else t)
        d, i = kz.kneighbors(150)
    od, oi = O.kiez_pipeline(s, None if single else t, 150, 150, metric, 2, None, {})
    np.testing.assert_array_equal(i, oi)
    np.testing.assert_allclose(d, od, rtol=1e-9, atol=1e-12)
    # (hubness reductions with more than 128 candidates: tests/test_gpu_merge.py; beyond the library's 4095 neighbours it says so)
    with pytest.raises(NotImplementedError, match="exceeds"):
        Kiez(n_candidates=5000, algorithm="SklearnNN", hubness="CSLS")
