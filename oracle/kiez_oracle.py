"""CPU oracle for the kiez exact-kNN + hubness-reduction hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm (dobraczka/kiez v0.5.0, whose
arithmetic lives in scikit-learn 1.7.2 / numpy 2.2 / scipy 1.15 as installed in this image;
the reference pins sklearn 1.3.2 / numpy 1.24.4, `poetry.lock`).  It is the *checker* for the
HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product package ``kiez_amd`` never imports anything from ``oracle/``.

Parity pin: every function here is checked against golden vectors produced by importing the real
reference in the build container (``tools/gen_golden.py`` -> ``tests/golden/*.npz``; test:
``tests/test_oracle_golden.py``).  The reference's own test-suite pins no numeric value of this path
(SURVEY.md §4), so those generated vectors are the pin.

Conventions shared with the HIP path (documented in DESIGN.md):
* float32 inputs are treated as their exact float64 casts.  For the euclidean family that is what
  scikit-learn does itself (ArgKmin upcasts chunks to float64); for cosine it removes the dependence
  of the neighbour order on sgemm rounding (SURVEY.md §8c caution 2).
* exact distance ties are ordered by smaller index (sklearn's heap order on exact duplicates is
  implementation-defined, `sklearn/utils/_sorting.pyx:39-42`).
"""
from __future__ import annotations

import numpy as np
from scipy import special

METRICS = ("euclidean", "sqeuclidean", "cosine", "manhattan", "chebyshev")


def canonical_metric(metric: str, p: float = 2) -> str:
    """One name per distinct metric.  minkowski(p=2) == euclidean: kiez/neighbors/exact/sklearn_nearest_neighbors.py:51-65
    (defaults); scikit-learn's aliases: `sklearn/metrics/_dist_metrics.pyx.tp` (METRIC_MAPPING: l2 = euclidean, l1 = cityblock =
    manhattan; DistanceMetric.get_metric: minkowski with p = 1 / 2 / inf -> Manhattan / Euclidean / Chebyshev).  Any other
    minkowski exponent p >= 1 is written 'minkowski[p]'."""
    if metric.startswith("minkowski["):
        return metric
    if metric == "minkowski":
        if not p >= 1:
            raise ValueError("minkowski needs p >= 1")
        if p == 2:
            return "euclidean"
        if p == 1:
            return "manhattan"
        if np.isinf(p):
            return "chebyshev"
        return f"minkowski[{float(p)!r}]"
    if metric in ("l2",):
        return "euclidean"
    if metric in ("l1", "cityblock"):
        return "manhattan"
    if metric not in METRICS:
        raise ValueError(f"unsupported metric {metric}")
    return metric


def minkowski_family_rdist(q: np.ndarray, y: np.ndarray, metric: str) -> np.ndarray:
    """Reduced distances [len(q), len(y)] of manhattan / chebyshev / 'minkowski[p]' as scikit-learn's generic DistanceMetric
    computes them for the brute-force search (`sklearn/metrics/_dist_metrics.pyx.tp`: ManhattanDistance.dist,
    ChebyshevDistance.dist, MinkowskiDistance.rdist; DistanceMetric32 for float32 inputs): the difference x_j - y_j in the INPUT
    dtype, |.| (to the power p) accumulated in float64 in feature order, the result rounded to the input dtype.
    q and y must have the same dtype (float32 or float64)."""
    assert q.dtype == y.dtype and q.dtype in (np.float32, np.float64)
    p = float(metric[len("minkowski["):-1]) if metric.startswith("minkowski[") else None
    acc = np.zeros((q.shape[0], y.shape[0]), dtype=np.float64)
    for j in range(q.shape[1]):
        df = np.abs(q[:, j, None] - y[None, :, j]).astype(np.float64)      # (the subtraction in the input dtype)
        if metric == "chebyshev":
            np.maximum(acc, df, out=acc)
        elif metric == "manhattan":
            acc += df
        else:
            acc += df ** p
    if q.dtype == np.float32:
        acc = acc.astype(np.float32).astype(np.float64)
    return acc


# --------------------------------------------------------------------------------------------
# a-3: brute-force kNN  (kiez/neighbors/exact/sklearn_nearest_neighbors.py:96-101 ->
#       sklearn/neighbors/_base.py:862-915)
# --------------------------------------------------------------------------------------------
def _row_sqnorms(x64: np.ndarray) -> np.ndarray:
    return np.einsum("ij,ij->i", x64, x64)


def _topk_rows(val: np.ndarray, k: int):
    """k smallest per row, ascending by (value, index)."""
    n = val.shape[1]
    if k < n:
        part = np.argpartition(val, k - 1, axis=1)[:, :k].astype(np.int64)
        pv = np.take_along_axis(val, part, axis=1)
        order = np.lexsort((part, pv), axis=1)
        out_i = np.take_along_axis(part, order, axis=1)
        # rows where the k-th value is tied with an element left outside: resolve ties by index
        kth = pv.max(axis=1)
        tied = np.flatnonzero((val <= kth[:, None]).sum(axis=1) > k)
        for r in tied:
            cand = np.flatnonzero(val[r] <= kth[r])
            o = np.lexsort((cand, val[r, cand]))[:k]
            out_i[r] = cand[o]
    else:
        idx = np.broadcast_to(np.arange(n), val.shape)
        out_i = np.lexsort((idx, val), axis=1).astype(np.int64)
    out_v = np.take_along_axis(val, out_i, axis=1)
    return out_v, out_i


def knn_exact(query, index, k: int, metric: str = "euclidean", exclude_self: bool = False,
              chunk_bytes: int = 1 << 28, threads: int = 0):
    """Exact k nearest index rows for every query row, sorted ascending.

    euclidean family — sklearn `EuclideanArgKmin`
    (`sklearn/metrics/_pairwise_distances_reduction/_argkmin.pyx.tp:311-510`): float64
    d2 = |x|^2 - 2 x.y + |y|^2, clamp at 0, sqrt at the end (skipped for sqeuclidean).
    cosine — `sklearn/metrics/pairwise.py:1166-1175,1728-1736`: normalise rows, S = Xn Yn^T,
    1 - S, clip to [0, 2]; then the k smallest (`sklearn/neighbors/_base.py:749-753`).
    manhattan / chebyshev / minkowski[p] — the generic `ArgKmin` over a DatasetsPair of the metric object
    (`_argkmin.pyx.tp:170-310` with `minkowski_family_rdist` above), ranked by the reduced distance.
    exclude_self — `sklearn/neighbors/_base.py:828-834,937-965`: ask for k+1, drop the entry whose
    index equals the row id (or the first entry when it is absent).
    Returns float64 distances and int64 indices.  For float32 query AND index with metric euclidean the
    float64 values are float32-representable (see the sqrt step below) — measured on sklearn 1.7.2.
    threads > 1: the row chunks are worked on by that many Python threads (numpy's partition and BLAS release
    the GIL; every chunk is computed exactly as in the serial loop, so the result does not depend on it) —
    for the full-size checks of tests/ on a many-core host.
    """
    metric = canonical_metric(metric)
    both_f32 = np.asarray(query).dtype == np.float32 and np.asarray(index).dtype == np.float32
    family = metric in ("manhattan", "chebyshev") or metric.startswith("minkowski[")
    if family:
        # (no float64 cast of float32 inputs here: DistanceMetric32 subtracts in float32)
        dt = np.float32 if both_f32 else np.float64
        q_in = np.ascontiguousarray(query, dtype=dt)
        y_in = np.ascontiguousarray(index, dtype=dt)
    q = np.ascontiguousarray(query, dtype=np.float64)
    y = np.ascontiguousarray(index, dtype=np.float64)
    n_q, n_i = q.shape[0], y.shape[0]
    kk = k + 1 if exclude_self else k
    if kk > n_i:
        raise ValueError(
            f"Expected n_neighbors {'<' if exclude_self else '<='} n_samples_fit, but n_neighbors = {k}, "
            f"n_samples_fit = {n_i}, n_samples = {n_q}")
    if metric == "cosine":
        qn = np.sqrt(_row_sqnorms(q))
        yn = np.sqrt(_row_sqnorms(y))
        qn[qn == 0.0] = 1.0
        yn[yn == 0.0] = 1.0
        q = q / qn[:, None]
        y = y / yn[:, None]
    elif not family:
        qsq = _row_sqnorms(q)
        ysq = _row_sqnorms(y)
    rows = max(1, int(chunk_bytes // (8 * max(n_i, 1))))
    dist = np.empty((n_q, kk), dtype=np.float64)
    ind = np.empty((n_q, kk), dtype=np.int64)
    def chunk(s):
        e = min(n_q, s + rows)
        if family:
            dist[s:e], ind[s:e] = _topk_rows(minkowski_family_rdist(q_in[s:e], y_in, metric), kk)
            return
        g = q[s:e] @ y.T
        if metric == "cosine":
            g *= -1.0
            g += 1.0
            np.clip(g, 0.0, 2.0, out=g)
        else:
            g *= -2.0
            g += qsq[s:e, None]
            g += ysq[None, :]
            np.maximum(g, 0.0, out=g)
        dist[s:e], ind[s:e] = _topk_rows(g, kk)

    starts = range(0, n_q, rows)
    if threads > 1 and len(starts) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(chunk, starts))
    else:
        for s in starts:
            chunk(s)
    if metric == "euclidean":
        if both_f32:
            # ArgKmin32 orders in float64 but converts the surrogate with the float32 metric object:
            # dist = (double) sqrtf((float) d2)  (`_argkmin.pyx.tp:285-295`, `_dist_metrics.pyx.tp:1018-1019`
            # with INPUT_DTYPE_t = float32).  sqeuclidean skips this step (`_argkmin.pyx.tp:393-396`).
            dist = np.sqrt(dist.astype(np.float32)).astype(np.float64)
        else:
            np.sqrt(dist, out=dist)
    elif metric.startswith("minkowski["):
        # MinkowskiDistance._rdist_to_dist: rdist ** (1 / p), in the input dtype's metric object: float32 inputs give
        # float32-representable distances (measured on scikit-learn 1.7.2: (float32) pow((double) rdist, 1 / p) reproduces all)
        dist = dist ** (1.0 / float(metric[len("minkowski["):-1]))
        if both_f32:
            dist = dist.astype(np.float32).astype(np.float64)
    if exclude_self:
        rows_id = np.arange(n_q)[:, None]
        mask = ind != rows_id
        none_self = mask.all(axis=1)
        mask[none_self, 0] = False
        ind = ind[mask].reshape(n_q, k)
        dist = dist[mask].reshape(n_q, k)
    return dist, ind


# --------------------------------------------------------------------------------------------
# a-6: final candidate sort  (kiez/hubness_reduction/base.py:72-87, numpy branch :81-86)
# --------------------------------------------------------------------------------------------
def sort_topk(dist: np.ndarray, ind: np.ndarray, k: int):
    """`np.argpartition(dist, kth=arange(k))[:, :k]` + take_along_axis, restated as what it is for
    k >= 2 (measured on numpy 2.2): a selection sort — for i in 0..k-1 pick the
    FIRST strict minimum in positions [i, K) and SWAP it into position i.  For k == 1 the same rule
    (first minimum) is the scalar-path behaviour."""
    d = np.array(dist, copy=True)
    i_ = np.array(ind, copy=True)
    n, K = d.shape
    rows = np.arange(n)
    for i in range(min(k, K)):
        p = i + np.argmin(d[:, i:], axis=1)  # argmin returns the first minimum
        di, dp = d[rows, i].copy(), d[rows, p].copy()
        d[rows, i], d[rows, p] = dp, di
        ii, ip = i_[rows, i].copy(), i_[rows, p].copy()
        i_[rows, i], i_[rows, p] = ip, ii
    return d[:, :k], i_[:, :k]


# --------------------------------------------------------------------------------------------
# a-8 CSLS  (kiez/hubness_reduction/csls.py:53-54, 85-96)
# --------------------------------------------------------------------------------------------
def csls_transform(dist, ind, dist_t2s):
    r_train = dist_t2s.mean(axis=1)                     # csls.py:90
    r_test = dist.mean(axis=1).reshape(-1, 1)           # csls.py:91
    return 2 * dist - r_test - r_train[ind]             # csls.py:93


# --------------------------------------------------------------------------------------------
# a-10 Local scaling  (kiez/hubness_reduction/local_scaling.py:82-83, 129-151)
# --------------------------------------------------------------------------------------------
def ls_transform(dist, ind, dist_t2s, method="standard"):
    method = method.lower()
    if method in ("ls", "standard"):
        r_t = dist_t2s[:, -1]                                   # :136
        r_s = dist[:, -1].reshape(-1, 1)                        # :137
        inner = -1 * dist**2 / (r_s * r_t[ind])                 # :138
        return 1.0 - np.exp(inner)                              # :139-140
    if method == "nicdm":
        r_t = dist_t2s.mean(axis=1)                             # :143
        r_s = dist.mean(axis=1).reshape(-1, 1)                  # :144
        return dist / np.sqrt(r_s * r_t[ind])                   # :145-147
    raise ValueError(f"Internal: Invalid method {method}. Try 'ls' or 'nicdm'.")


# --------------------------------------------------------------------------------------------
# a-9 Mutual proximity  (kiez/hubness_reduction/mutual_proximity.py:92-104, 166-212)
# --------------------------------------------------------------------------------------------
def _norm_sf(x, loc, scale):
    """scipy.stats.norm.sf(x, loc, scale) == ndtr(-(x-loc)/scale)  (mutual_proximity.py:179-182)."""
    return special.ndtr(-((x - loc) / scale))


def mp_normal_transform(dist, ind, dist_t2s):
    mu_t = np.nanmean(dist_t2s, axis=1)                 # :102
    sd_t = np.nanstd(dist_t2s, axis=1)                  # :103
    mu = np.nanmean(dist, axis=1).reshape(-1, 1)        # :177
    sd = np.nanstd(dist, axis=1).reshape(-1, 1)         # :178
    with np.errstate(divide="ignore", invalid="ignore"):
        p1 = _norm_sf(dist, mu, sd)                     # :179
        p2 = _norm_sf(dist, mu_t[ind], sd_t[ind])       # :180-182
    return 1 - p1 * p2                                  # :183


def mp_empiric_transform(dist, ind, dist_t2s, ind_t2s):
    """mutual_proximity.py:185-212.  For query i, candidate j:
    out[i,j] = 1 - #{m : d[i,m] > d[i,j] and T_j[m] > d[i,j]} / K, where
    T_j[m] = dist_t2s[c_j, p] if ind_t2s[c_j, p] == c_m for some p (target id c_m is looked up in a
    list of SOURCE ids: reference behaviour, kept) else dist_t2s[c_j, K-1] + 1e-6."""
    n, K = dist.shape
    out = np.empty_like(dist)
    for i in range(n):
        c = ind[i]
        d = dist[i]
        rows_i = ind_t2s[c]                 # [K, Kt]
        rows_d = dist_t2s[c]                # [K, Kt]
        fill = rows_d[:, -1] + 1e-6         # [K]
        T = np.repeat(fill[:, None], K, axis=1).astype(np.float64)  # T[j, m]
        # matches[j, p, m] = ind_t2s[c_j, p] == c_m ; a kNN row holds distinct ids -> at most one p
        match = rows_i[:, :, None] == c[None, None, :]
        j_idx, p_idx, m_idx = np.nonzero(match)
        T[j_idx, m_idx] = rows_d[j_idx, p_idx]
        cnt = ((d[None, :] > d[:, None]) & (T > d[:, None])).sum(axis=1)
        out[i] = 1.0 - cnt / K
    return out


# --------------------------------------------------------------------------------------------
# a-11 DisSimLocal  (kiez/hubness_reduction/dis_sim.py:96-107, 139-181)
# --------------------------------------------------------------------------------------------
def dsl_fit(ind_t2s, source, target):
    centroids = source[ind_t2s].mean(axis=1)                        # :96
    diff = target - centroids
    return np.einsum("ij,ij->i", diff, diff)                        # :102 (row_norms squared)


def dsl_transform(dist, ind, query, target, t2c, squared: bool):
    """dis_sim.py:139-181 evaluated in float64 (for float32 inputs the reference mixes float32
    einsum/mean results whose rounding is CPU-feature dependent; DESIGN.md 'DSL numerics')."""
    q = np.asarray(query, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    nb = t[ind]                                                     # [n, K, d]
    diff = q[:, None, :] - nb
    hub = np.einsum("nkd,nkd->nk", diff, diff)                      # :153-157 squared euclidean
    cent = nb.mean(axis=1)                                          # :159
    smc = q - cent
    s2c = (smc**2).sum(axis=1)                                      # :160-162
    hub = hub - s2c.reshape(-1, 1)                                  # :165
    hub = hub - t2c[ind]                                            # :166
    mn = hub.min()                                                  # :171
    if mn < 0.0:                                                    # :172 (_MINIMUM_DIST = 0.0)
        hub = hub + (-mn)
    if not squared:
        hub = hub ** (1 / 2)                                        # :176-177
    return hub


# --------------------------------------------------------------------------------------------
# a-4/a-5/a-7: the fit + kneighbors orchestration  (kiez/hubness_reduction/base.py:33-50, 89-122;
#              kiez/neighbors/neighbor_algorithm_base.py:53-136)
# --------------------------------------------------------------------------------------------
def kiez_pipeline(source, target=None, n_candidates=10, k=None, metric="euclidean", p=2,
                  hubness=None, hubness_kwargs=None, return_intermediates=False, query_rows=None):
    """Oracle for `Kiez(n_candidates, 'SklearnNN', {'metric':..}, hubness, hubness_kwargs).fit(s, t).kneighbors(k)`.

    query_rows=n (two-source mode, not DSL): evaluate the forward pass / transform for the first n source rows only
    (the fit state still uses ALL source rows) — used to spot-check full-size runs."""
    hubness_kwargs = dict(hubness_kwargs or {})
    metric_c = canonical_metric(metric, p)
    single = target is None
    tgt = source if single else target
    K = n_candidates
    if k is None or k > K:
        k = K
    hub = None if hubness is None else str(hubness).lower()
    if hub in (None, "no", "nohubnessreduction"):
        kk = min(k, tgt.shape[0])
        d, i = knn_exact(source if query_rows is None else source[:query_rows], tgt, kk, metric_c,
                         exclude_self=single)   # base.py:120-122
        return (d, i) if not return_intermediates else (d, i, {})
    # reverse pass: explicit query=target, so self is NOT stripped even for a single source
    # (neighbor_algorithm_base.py:119; base.py:37-42)
    Kr = min(K, source.shape[0])
    dist_t2s, ind_t2s = knn_exact(tgt, source, Kr, metric_c, exclude_self=False)
    Kf = min(K, tgt.shape[0])
    fwd_src = source
    if query_rows is not None:
        if single or hub in ("dissimlocal", "dsl"):
            raise ValueError("query_rows needs two-source mode and a row-local transform")
        fwd_src = source[:query_rows]
    dist_s2t, ind_s2t = knn_exact(fwd_src, tgt, Kf, metric_c, exclude_self=single)   # base.py:92-94
    if hub == "csls":
        hr = csls_transform(dist_s2t, ind_s2t, dist_t2s)
    elif hub in ("localscaling", "ls"):
        hr = ls_transform(dist_s2t, ind_s2t, dist_t2s, hubness_kwargs.get("method", "standard"))
    elif hub in ("mutualproximity", "mp"):
        method = hubness_kwargs.get("method", "normal")
        if method in ("exact", "empiric"):
            hr = mp_empiric_transform(dist_s2t, ind_s2t, dist_t2s, ind_t2s)
        elif method in ("normal", "gaussi"):
            hr = mp_normal_transform(dist_s2t, ind_s2t, dist_t2s)
        else:
            raise ValueError(f'Mutual proximity method "{method}" not recognized.')
    elif hub in ("dissimlocal", "dsl"):
        if metric_c not in ("euclidean", "sqeuclidean"):                      # dis_sim.py:47-61 (minkowski only with p = 2)
            raise ValueError("DisSimLocal only supports squared Euclidean distances")
        squared = metric_c == "sqeuclidean"                                  # dis_sim.py:47-56
        s64 = np.asarray(source, dtype=np.float64)
        t64 = np.asarray(tgt, dtype=np.float64)
        t2c = dsl_fit(ind_t2s, s64, t64)
        hr = dsl_transform(dist_s2t, ind_s2t, s64, t64, t2c, squared)
    else:
        raise ValueError(f"unknown hubness reduction {hubness}")
    d, i = sort_topk(hr, ind_s2t, k)                                         # base.py:103-105
    if return_intermediates:
        return d, i, dict(dist_t2s=dist_t2s, ind_t2s=ind_t2s, dist_s2t=dist_s2t, ind_s2t=ind_s2t,
                          transformed=hr)
    return d, i
