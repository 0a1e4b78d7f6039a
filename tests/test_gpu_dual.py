"""Shared sweep (kz_knn_dual): both search directions out of one pass over the distance matrix must be IDENTICAL -- indices
and distances, bit for bit -- to two ordinary kz_knn searches, on every shape, metric, dtype and list length, through the
fallback routes (event-buffer overflow, poisoned or overflowing log, certification failures, ineligible settings) and
through the API layers built on it.  Needs an MI355X: `pytest -m gpu`."""
import os
import warnings

import numpy as np
import pytest

from tests.golden_util import HUB, case_params, knife_edge_rows, knife_edge_topk_ok, ktag, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    c = N.Context.get()
    c.set_option("dual_force", 1)   # the test shapes are far below the size at which the shared sweep pays
    yield c
    for name, value in (("dual_force", 0), ("dual_stride", 1), ("chunk_rows", 0), ("eps_scale", 1.0), ("precision", 0), ("dual_max_gb", 0),
                        ("dual_overlap", 1), ("dual_sample_short", 1), ("dual_short_main", 1), ("dual_short_min_tiles", 128), ("dual_rev_long", 1), ("dual_short_extra", 48), ("esc_bf", 1),
                        ("dual_rank", 0)):
        c.set_option(name, value)


def _data(kind, n, d, seed, dtype):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, d)).astype(dtype)
    if kind == "normal":
        return rng.standard_normal((n, d)).astype(dtype)
    if kind == "clustered":   # strong hubness: a few dense clusters + a sparse cloud, rows in cluster order
        centres = rng.standard_normal((8, d)) * 3
        sizes = rng.multinomial(n - n // 5, np.ones(8) / 8)
        parts = [centres[c] + 0.2 * rng.standard_normal((sizes[c], d)) for c in range(8)]
        parts.append(5 * rng.standard_normal((n - sum(sizes), d)))
        return np.concatenate(parts).astype(dtype)
    if kind == "duplicates":  # exact ties: every row occurs several times
        base = rng.random((max(n // 7, 8), d))
        return base[rng.integers(0, len(base), n)].astype(dtype)
    raise ValueError(kind)


def _both_ways(ctx, a, b, k, metric):
    from kiez_amd import _native as N
    am, bm = N.DeviceMatrix(ctx, a, metric), N.DeviceMatrix(ctx, b, metric)
    ctx.set_option("dual_force", 0)
    d_ab, i_ab, _ = N.knn(ctx, am, bm, k)
    d_ba, i_ba, _ = N.knn(ctx, bm, am, k)
    ctx.set_option("dual_force", 1)
    (xd, xi, s_ab), (yd, yi, s_ba) = N.knn_dual(ctx, am, bm, k)
    return (d_ab.numpy(), i_ab.numpy(), d_ba.numpy(), i_ba.numpy()), (xd.numpy(), xi.numpy(), yd.numpy(), yi.numpy()), s_ab, s_ba


def _oracle_sample(a, b, k, metric, dual):
    """... and not only against the library's own two searches: a row sample of BOTH directions against the oracle."""
    from oracle import kiez_oracle as O
    a64, b64 = (a.astype(np.float64), b.astype(np.float64)) if metric == "cosine" else (a, b)
    ra, rb = np.arange(0, len(a), max(1, len(a) // 200))[:200], np.arange(0, len(b), max(1, len(b) // 200))[:200]
    np.testing.assert_array_equal(dual[1][ra], O.knn_exact(a64[ra], b64, k, metric)[1])
    np.testing.assert_array_equal(dual[3][rb], O.knn_exact(b64[rb], a64, k, metric)[1])


def _assert_same(sep, dual):
    for name, x, y in zip(("dist a->b", "ind a->b", "dist b->a", "ind b->a"), sep, dual):
        np.testing.assert_array_equal(y, x, err_msg=name)


@pytest.mark.parametrize("kind,na,nb,d,k,metric,dtype", [
    ("uniform", 20000, 6000, 64, 10, "euclidean", np.float32),      # K' = 16, three workgroups per CU
    ("uniform", 9000, 30011, 72, 5, "sqeuclidean", np.float64),      # odd slice count, a smaller than b, float64
    ("normal", 12000, 8000, 200, 50, "cosine", np.float32),          # K' = 64, 13 slices, two workgroups per CU
    ("uniform", 15000, 9000, 48, 26, "euclidean", np.float32),       # K' = 32
    ("normal", 12000, 5000, 32, 100, "euclidean", np.float32),       # K' = 128
    ("uniform", 16384, 4096, 200, 10, "euclidean", np.float32),      # 13 slices at three workgroups per CU (single fragment set)
    ("clustered", 20000, 12000, 40, 10, "euclidean", np.float32),    # hubs: uneven event counts, rows in cluster order
    ("duplicates", 10000, 7000, 24, 10, "sqeuclidean", np.float32),  # exact ties in both directions
    ("uniform", 5000, 1029, 300, 3, "cosine", np.float64),           # 19 slices, ragged last tiles
])
def test_both_directions_identical_to_two_searches(ctx, kind, na, nb, d, k, metric, dtype):
    a, b = _data(kind, na, d, 1, dtype), _data(kind, nb, d, 2, dtype)
    sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, k, metric)
    _assert_same(sep, dual)
    if kind != "duplicates":
        _oracle_sample(a, b, k, metric, dual)
    if kind != "duplicates":                      # (exact ties: more than a quarter of the rows fail the fp16 certification, the
        assert s_ab["dual"] == 1                  #  call gives up on sharing; the forward direction otherwise always shares ...
    assert s_ab["max_err_ratio"] < 1.0 and s_ba["max_err_ratio"] < 1.0
    if s_ba["dual"] == 1:                         # ... the reverse one unless the log overflowed (tiny inputs: many events per tile)
        # (about rank x stride events per row, and the automatic rank sits near k / stride: ~k per row on average -- the rows that
        #  get fewer than k are searched again, which the identity above has just checked)
        assert s_ba["n_events"] >= 0.8 * k * nb
        assert s_ba["n_logged_groups"] * 4 >= s_ba["n_events"]


def test_reverse_direction_uses_the_sweep_on_a_shape_where_it_pays(ctx):
    """250k query rows: events are rare per tile, nothing overflows, (nearly) every row of b is certified from its events."""
    a, b = _data("uniform", 250000, 64, 3, np.float32), _data("uniform", 20000, 64, 4, np.float32)
    sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, 10, "euclidean")
    _assert_same(sep, dual)
    assert s_ab["dual"] == 1 and s_ba["dual"] == 1
    assert 4 * 8 < s_ba["n_events"] / len(b) < 40 * 10      # rank x stride: rank chosen within [8, k + 1], stride within [4, 32]
    assert s_ba["n_logged_groups"] < 1.3 * s_ba["n_events"]   # the per-tile threshold (rows sorted by threshold) is nearly exact
    assert s_ba["n_escalated_rows"] < 0.01 * len(b)


@pytest.mark.parametrize("stride", [2, 5, 32])
def test_sample_strides(ctx, stride):
    ctx.set_option("dual_stride", stride)
    a, b = _data("uniform", 40000, 32, 5, np.float32), _data("uniform", 3000, 32, 6, np.float32)
    sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, 10, "euclidean")
    _assert_same(sep, dual)


def test_query_side_swept_in_several_chunks(ctx):
    """Events of a row of b accumulate over the chunks of a."""
    ctx.set_option("chunk_rows", 4096)
    a, b = _data("uniform", 30000, 64, 7, np.float32), _data("uniform", 5000, 64, 8, np.float32)
    sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, 10, "euclidean")
    _assert_same(sep, dual)
    assert s_ab["dual"] == 1


def test_rows_that_fail_certification_are_searched_again(ctx):
    """A huge rounding bound: no candidate set is certified in either direction, every row goes down the tiers."""
    ctx.set_option("eps_scale", 1e12)
    a, b = _data("uniform", 6000, 32, 9, np.float32), _data("uniform", 2000, 32, 10, np.float32)
    sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, 5, "euclidean")
    _assert_same(sep, dual)
    assert s_ba["n_escalated_rows"] + s_ba["n_fallback_rows"] >= len(b) or s_ba["dual"] == 0


def test_settings_without_a_shared_sweep_fall_back_to_two_searches(ctx):
    from kiez_amd import _native as N
    a, b = _data("uniform", 5000, 32, 11, np.float32), _data("uniform", 3000, 32, 12, np.float32)
    am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
    ref = N.knn(ctx, am, bm, 7)[:2], N.knn(ctx, bm, am, 7)[:2]
    for name, value in (("dual_stride", 0), ("precision", 1), ("precision", 2), ("dual_force", 0)):
        ctx.set_option(name, value)
        (xd, xi, s_ab), (yd, yi, s_ba) = N.knn_dual(ctx, am, bm, 7)
        assert s_ab["dual"] == 0 and s_ba["dual"] == 0, name
        np.testing.assert_array_equal(xi.numpy(), ref[0][1].numpy())
        np.testing.assert_array_equal(yd.numpy(), ref[1][0].numpy())
        ctx.set_option("dual_stride", 1)
        ctx.set_option("precision", 0)
        ctx.set_option("dual_force", 1)
    # more than 110 neighbours: exact-only route in both directions
    (xd, xi, s_ab), (yd, yi, s_ba) = N.knn_dual(ctx, am, bm, 120)
    rd, ri, _ = N.knn(ctx, bm, am, 120)
    np.testing.assert_array_equal(yi.numpy(), ri.numpy())
    with pytest.raises(ValueError):
        N.knn_dual(ctx, am, am, 5)
    with pytest.raises(ValueError):
        N.knn_dual(ctx, am, bm, 3001)


@pytest.mark.parametrize("hub,kw", [("CSLS", {}), ("LocalScaling", {"method": "nicdm"}), ("LocalScaling", {"method": "standard"}),
                                    ("MutualProximity", {"method": "normal"}), ("MutualProximity", {"method": "empiric"}),
                                    ("DisSimLocal", {})])
@pytest.mark.parametrize("swap", [False, True])
def test_kiez_api_same_result_with_and_without_the_shared_sweep(ctx, hub, kw, swap):
    from kiez_amd import Kiez
    s, t = _data("uniform", 30000, 48, 13, np.float32), _data("uniform", 20000, 48, 14, np.float32)
    if swap:
        s, t = t, s
    out = []
    for shared in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kz = Kiez(n_candidates=10, algorithm="SklearnNN", algorithm_kwargs={"metric": "euclidean"}, hubness=hub,
                      hubness_kwargs=dict(kw))
            kz.hubness._shared_sweep = shared
            out.append(kz.fit(s, t).kneighbors(5))
            if shared:
                assert kz.algorithm.last_stats["dual"] == 1
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][0], out[1][0])


@pytest.mark.parametrize("case,tag,k", [p for p in case_params() if p[0] in ("c0_two_source", "f32_euclidean", "cosine_k50")])
def test_golden_pipeline_through_the_shared_sweep(ctx, case, tag, k):
    """The reference's own outputs (tests/golden/), with the shared sweep forced on these small cases."""
    from kiez_amd import Kiez
    g = load_case(case)
    hname, kw = HUB[tag]
    if hname in (None, "NoHubnessReduction") or g["_target"] is None:
        pytest.skip("no reverse search in this case")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=g["_K"], algorithm="SklearnNN", algorithm_kwargs=dict(metric=g["_metric"], p=g["_p"]),
                  hubness=hname, hubness_kwargs=dict(kw))
        kz.fit(g["source"], g["_target"])
        d, i = kz.kneighbors(k)
    ref_d, ref_i = g[f"{tag}__k{ktag(k)}__dist"], g[f"{tag}__k{ktag(k)}__ind"]
    keep = np.ones(len(i), dtype=bool)
    if tag == "mp_empiric":
        keep &= ~knife_edge_rows(g["mp_empiric__ind_s2t"])
        for r in np.flatnonzero(~keep):
            assert knife_edge_topk_ok(ref_d[r], ref_i[r], d[r], i[r], r, g["_K"], g["mp_empiric__ind_t2s"]), f"knife-edge row {r}"
    np.testing.assert_array_equal(i[keep], ref_i[keep])
    np.testing.assert_allclose(d[keep], ref_d[keep], rtol=1e-5, atol=1e-6)


SHARDED_SCRIPT = r"""
import os, sys, warnings
sys.path.insert(0, %r)
os.environ["KIEZ_AMD_WITH_TORCH"] = "1"
import torch
import numpy as np
from kiez_amd.distributed import Comm, HipEngine, ShardedKiez
warnings.simplefilter("ignore")
eng = HipEngine(0)
eng.ctx.set_option("dual_force", 1)
rng = np.random.default_rng(15)
s, t = rng.random((12000, 32), dtype=np.float32), rng.random((30000, 32), dtype=np.float32)
for hub, kw in (("CSLS", {}), ("LocalScaling", {"method": "nicdm"}), ("MutualProximity", {"method": "normal"}),
                ("MutualProximity", {"method": "empiric"}), ("DisSimLocal", {})):
    res = []
    for shared in (True, False):
        sk = ShardedKiez(n_candidates=10, algorithm_kwargs={"metric": "euclidean"}, hubness=hub,
                         hubness_kwargs=dict(kw, shared_sweep=shared), engine=eng, comm=Comm())
        sk.fit(eng.to_engine(s), eng.to_engine(t))
        assert sk.shared == shared, (hub, shared, sk.shared)
        if shared:
            assert eng.last_stats["dual"] == 1 and eng.last_stats_reverse["dual"] == 1, (eng.last_stats, eng.last_stats_reverse)
        d, i = sk.kneighbors(5)
        torch.cuda.synchronize()
        res.append((d.cpu().numpy(), i.cpu().numpy()))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0], res[1][0]), hub
print("SHARDED_DUAL_OK")
"""


def test_sharded_pipeline_single_rank():
    """ShardedKiez on the HIP engine (what bench.py runs): shared sweep against two searches, every hubness kind.
    Subprocess: torch has to be imported before libkiez_amd.so is loaded (one HIP runtime per process)."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-c", SHARDED_SCRIPT % str(root)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SHARDED_DUAL_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


def test_large_k_keeps_the_event_buffers_within_lds(ctx):
    """k = 100 with the largest stride the option admits: the stride is cut back so that a row's events still fit the
    select kernel's LDS."""
    ctx.set_option("dual_stride", 64)
    a, b = _data("normal", 20000, 24, 21, np.float32), _data("normal", 3000, 24, 22, np.float32)
    sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, 100, "euclidean")
    _assert_same(sep, dual)


@pytest.mark.parametrize("hub,kw", [("CSLS", {}), ("LocalScaling", {"method": "standard"}), ("MutualProximity", {"method": "normal"}),
                                    ("MutualProximity", {"method": "empiric"}), ("DisSimLocal", {})])
@pytest.mark.parametrize("kind,metric", [("uniform", "euclidean"), ("duplicates", "sqeuclidean")])
def test_single_source_one_search_serves_both_views(hub, kw, kind, metric):
    """fit(source) alone: the reverse pass (rows keep themselves) and the forward pass (rows stripped as sklearn does) are
    two views of ONE search for K + 1 neighbours (kz_split_self) -- same results as the two searches, also on exact
    duplicates (where a row is not necessarily its own first neighbour)."""
    from kiez_amd import Kiez
    s = _data(kind, 9000, 40, 31, np.float32)
    out = []
    for shared in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kz = Kiez(n_candidates=10, algorithm="SklearnNN", algorithm_kwargs={"metric": metric}, hubness=hub,
                      hubness_kwargs=dict(kw))
            kz.hubness._shared_sweep = shared
            out.append(kz.fit(s).kneighbors(5))
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][0], out[1][0])


@pytest.mark.parametrize("na,nb,d,k,metric,dtype", [(30000, 5000, 64, 1, "euclidean", np.float32), (30000, 5000, 64, 2, "cosine", np.float64),
                                                    (5000, 30000, 33, 1, "sqeuclidean", np.float32)])
def test_one_or_two_neighbours(ctx, na, nb, d, k, metric, dtype):
    """k = 1: the threshold sits at the SECOND best sample key, so the best row is strictly above it also when it is a sample
    row -- (nearly) every row is certified from its handful of events."""
    a, b = _data("uniform", na, d, 41, dtype), _data("uniform", nb, d, 42, dtype)
    sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, k, metric)
    _assert_same(sep, dual)
    assert s_ba["dual"] == 1 and s_ba["n_escalated_rows"] < 0.02 * nb


def test_reverse_rows_of_a_k64_search_that_fail_are_researched_with_a_valid_list_length(ctx):
    """Regression (found by tools/fuzz_dual.py): K' = 64 with a few percent of uncertified reverse rows (exact duplicates)
    asked the re-search for lists of 4 x 64 = 256 entries; list lengths end at 128."""
    a, b = _data("duplicates", 21114, 200, 51, np.float32), _data("duplicates", 32829, 200, 52, np.float32)
    ctx.set_option("dual_rev_long", 0)   # (reverse lists of K' = 64 as then; with the default 128 every row of this case is certified)
    try:
        sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, 54, "sqeuclidean")
    finally:
        ctx.set_option("dual_rev_long", 1)
    _assert_same(sep, dual)
    assert s_ba["n_escalated_rows"] > 0 and s_ba["list_len"] == 64
    sep2, dual2, _, s_ba2 = _both_ways(ctx, a, b, 54, "sqeuclidean")
    _assert_same(sep2, dual2)
    assert s_ba2["list_len"] == 128 and s_ba2["n_escalated_rows"] <= s_ba["n_escalated_rows"]


def test_footprint_gate_and_cache_trim(ctx):
    """The shared sweep prices its transient buffers first (event buffers, log, permuted images): over budget it searches
    twice -- same results.  kz_ctx_trim hands the cached buffers back to the driver; the next call simply allocates again."""
    from kiez_amd import _native as N
    a, b = _data("uniform", 30000, 64, 1, np.float32), _data("uniform", 9000, 64, 2, np.float32)
    am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
    (d1, i1, s1), (e1, j1, t1) = N.knn_dual(ctx, am, bm, 10)
    assert s1["dual"] == 1 and t1["dual"] == 1
    ctx.set_option("dual_max_gb", 0.001)
    (d2, i2, s2), (e2, j2, t2) = N.knn_dual(ctx, am, bm, 10)
    assert s2["dual"] == 0 and t2["dual"] == 0, (s2, t2)          # over budget: two ordinary searches
    ctx.set_option("dual_max_gb", 0)
    ctx.trim()
    (d3, i3, s3), (e3, j3, t3) = N.knn_dual(ctx, am, bm, 10)
    assert s3["dual"] == 1
    for x, y, z in ((i1, i2, i3), (d1, d2, d3), (j1, j2, j3), (e1, e2, e3)):
        np.testing.assert_array_equal(x.numpy(), y.numpy())
        np.testing.assert_array_equal(x.numpy(), z.numpy())


def test_second_stream_on_and_off_give_the_same_result(ctx):
    from kiez_amd import _native as N
    a, b = _data("clustered", 25000, 48, 3, np.float32), _data("clustered", 12000, 48, 4, np.float32)
    am, bm = N.DeviceMatrix(ctx, a, "cosine"), N.DeviceMatrix(ctx, b, "cosine")
    res = []
    for ovl in (1, 0, 1):
        ctx.set_option("dual_overlap", ovl)
        (d, i, s), (e, j, t) = N.knn_dual(ctx, am, bm, 26)
        assert s["dual"] == 1 and t["dual"] == 1
        res.append((d.numpy(), i.numpy(), e.numpy(), j.numpy()))
    for r in res[1:]:
        for x, y in zip(res[0], r):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("k", [26, 50, 100])
def test_short_sample_lists_on_cluster_ordered_rows(ctx, k):
    """The sample sweep keeps lists of 16 (32) over several parts of the sample whatever k is; the parts interleave the sample's
    tiles, so data stored cluster by cluster (all near rows of a query in ONE stretch of the index) must give the same
    thresholds -- hence about the same number of events -- as lists of K', and the same neighbours."""
    from kiez_amd import _native as N
    a, b = _data("clustered", 25000, 48, 3, np.float32), _data("clustered", 12000, 48, 4, np.float32)
    am, bm = N.DeviceMatrix(ctx, a, "cosine"), N.DeviceMatrix(ctx, b, "cosine")
    ctx.set_option("dual_force", 1)
    out = {}
    try:
        for short in (0, 1):
            ctx.set_option("dual_sample_short", short)
            (d, i, s), (e, j, t) = N.knn_dual(ctx, am, bm, k)
            assert s["dual"] == 1 and t["dual"] == 1, (short, s, t)
            out[short] = (d.numpy(), i.numpy(), e.numpy(), j.numpy(), t["n_events"], t["n_fallback_rows"] + t["n_escalated_rows"])
    finally:
        ctx.set_option("dual_sample_short", 1)
        ctx.set_option("dual_force", 0)
    for x, y in zip(out[0][:4], out[1][:4]):
        np.testing.assert_array_equal(x, y)
    assert out[1][4] <= 1.25 * out[0][4] + 1000, (out[0][4], out[1][4])      # events: thresholds as tight as with lists of K'
    assert out[1][5] <= out[0][5] + 20, (out[0][5], out[1][5])                # rows searched again


@pytest.mark.parametrize("kind,metric", [("uniform", "euclidean"), ("normal", "cosine"), ("clustered", "euclidean"), ("duplicates", "sqeuclidean")])
@pytest.mark.parametrize("k", [13, 26, 27, 50, 54, 100])
def test_short_list_route_of_the_main_sweep(ctx, kind, metric, k):
    """13 .. 54 neighbours: the main sweep keeps k / 5 lists of 16 per query over dealt index ranges instead of one list of 32 / 64
    (kz_knn.hip "SHORT-LIST ROUTE").  Forced onto small inputs (ranges of 3 tiles, where a query's near rows DO crowd into single
    ranges and rows go down the tiers): both directions must equal two ordinary searches, and a row sample the oracle."""
    from kiez_amd import _native as N
    a, b = _data(kind, 9000, 40, 7, np.float32), _data(kind, 7000 if k <= 54 else 12000, 40, 8, np.float32)
    ctx.set_option("dual_short_min_tiles", 3)
    try:
        ref, got, s_ab, s_ba = _both_ways(ctx, a, b, k, metric)
    finally:
        ctx.set_option("dual_short_min_tiles", 128)
        ctx.set_option("dual_force", 0)
    if kind in ("uniform", "normal"):   # (clustered rows at this size need the float32 operands, tied distances at the K'-th place
        # cannot be certified: there the pass hands over to two searches -- the results must be right all the same)
        assert s_ab["dual"] == 1 and s_ba["dual"] == 1, (s_ab, s_ba)
        assert (k + 4) // 5 - 1 <= s_ab["n_splits"] <= (k + 4) // 5, s_ab   # (ranges of whole tiles: 94 tiles in 20 ranges of 5 are 19)
    for x, y in zip(ref, got):
        np.testing.assert_array_equal(x, y)
    if kind != "duplicates":   # (exact duplicates: the order among equal distances is the library's, not scikit-learn's)
        _oracle_sample(a, b, k, metric, got)


@pytest.mark.parametrize("rank", [-1, 0, 3, 1])
def test_threshold_rank_below_k_plus_one(ctx, rank):
    """`dual_rank`: the event threshold of a row at a LOWER rank of a thinner sample.  The k rows are then no longer among the events by
    construction: rows that get fewer than k events must come back uncertified and be searched again -- identical results for every
    rank, down to rank 1 (where a good share of the rows falls short)."""
    ctx.set_option("dual_rank", rank)
    try:
        a, b = _data("uniform", 60000, 48, 21, np.float32), _data("uniform", 9000, 48, 22, np.float32)
        sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, 10, "euclidean")
        _assert_same(sep, dual)
        _oracle_sample(a, b, 10, "euclidean", dual)
        assert s_ba["dual"] == 1
        if rank == 1:
            assert s_ba["n_escalated_rows"] > 0      # rows short of events went down the ordinary way
    finally:
        ctx.set_option("dual_rank", 0)


@pytest.mark.parametrize("kind,na,nb,d,k,metric,dtype", [
    ("uniform", 60000, 9000, 48, 10, "euclidean", np.float32),        # one list of 16 per row
    ("clustered", 50000, 30000, 64, 50, "cosine", np.float32),        # cluster-ordered rows, short lists in both sample levels
    ("normal", 40000, 41000, 200, 26, "sqeuclidean", np.float64),     # 13 slices
    ("duplicates", 30000, 8000, 32, 12, "euclidean", np.float32),     # exact ties among the sample rows and across the sample's edge
])
def test_nested_sample_rows_are_swept_once_and_come_out_the_same(ctx, kind, na, nb, d, k, metric, dtype):
    """NESTED sample (kz_knn_dual.h, round 5): the sampled rows of a -- the first tiles of its dealt image -- are no longer rows of the
    main sweep: their forward neighbours come out of the sample sweep (a shared sweep itself: events of the sample rows, thresholds
    from a third, small sweep), and the sample-row events of every row of b are read off the sample sweep's lists.  Results with
    the option on and off, and from two ordinary searches, must be identical; both directions also against the oracle."""
    a, b = _data(kind, na, d, 31, dtype), _data(kind, nb, d, 32, dtype)
    ctx.set_option("dual_nested", 0)
    try:
        sep, classic, s0_ab, s0_ba = _both_ways(ctx, a, b, k, metric)
        ctx.set_option("dual_nested", 1)
        _, nested, s_ab, s_ba = _both_ways(ctx, a, b, k, metric)
    finally:
        ctx.set_option("dual_nested", 1)
    assert s_ab["dual"] == 1 and s_ba["dual"] == 1 and s0_ba["dual"] == 1
    _assert_same(sep, classic)
    _assert_same(sep, nested)
    if kind != "duplicates":
        _oracle_sample(a, b, k, metric, nested)
    assert max(s_ab["max_err_ratio"], s_ba["max_err_ratio"]) < 1.0
    # the main sweep no longer logs the sample rows' events: fewer logged groups than the classic sweep, same events filed in all
    assert s_ba["n_logged_groups"] < s0_ba["n_logged_groups"]


def test_nested_sample_with_a_tiny_event_threshold_rank(ctx):
    """rank 1 leaves many rows of BOTH levels short of k events: the sample rows' own chain and the main chain send them to the
    ordinary search; same results."""
    ctx.set_option("dual_rank", 1)
    try:
        a, b = _data("uniform", 60000, 48, 41, np.float32), _data("uniform", 20000, 48, 42, np.float32)
        sep, dual, s_ab, s_ba = _both_ways(ctx, a, b, 10, "euclidean")
        _assert_same(sep, dual)
        assert s_ab["n_escalated_rows"] > 0 and s_ba["n_escalated_rows"] > 0
    finally:
        ctx.set_option("dual_rank", 0)
