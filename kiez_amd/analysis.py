"""Hubness estimation from a neighbour index matrix — `kiez.analysis.hubness_score`
(kiez/analysis/estimation.py:197-351) with the k-occurrence histogram and its reductions on the GPU.

The neighbour matrix is exactly what `Kiez.kneighbors` produces; it may be passed as a numpy array or as a device
array (`Kiez.kneighbors_device`).  Scalar post-processing of the handful of reduction results happens on the host.
"""
from __future__ import annotations

import ctypes as C
import math
import warnings
from typing import Optional, Union

import numpy as np

from . import _native as N

VALID_HUBNESS_MEASURES = [
    "all", "all_but_gini", "k_skewness", "k_skewness_truncnorm", "atkinson", "gini", "robinhood", "antihubs",
    "antihub_occurrence", "hubs", "hub_occurrence", "groupie_ratio", "k_occurrence",
]


def _truncnorm_third_moment(a: float) -> float:
    """scipy.stats.truncnorm(a, b).moment(3) for b -> +inf (estimation.py:52-57: b = (int64 max - mean) / std):
    m3 = 2*m1 + a^2 phi(a) / Z, m1 = phi(a) / Z, Z = 1 - Phi(a)."""
    phi = math.exp(-0.5 * a * a) / math.sqrt(2.0 * math.pi)
    z = 0.5 * math.erfc(a / math.sqrt(2.0))
    m1 = phi / z
    return 2.0 * m1 + a * a * phi / z


def hubness_score(nn_ind, target_samples: int, *, k: Optional[int] = None, hub_size: float = 2.0, verbose: int = 0,
                  return_value: str = "all_but_gini", store_k_occurrence: bool = False,
                  ctx: Optional[N.Context] = None) -> Union[float, dict]:
    """Same arguments and return values as the reference's `hubness_score`."""
    if isinstance(nn_ind, N.DeviceArray):
        ctx = nn_ind.ctx
        ind = nn_ind
        if ind.dtype != np.int64 or len(ind.shape) != 2:
            raise ValueError("device neighbour matrix must be a 2D int64 array")
    else:
        arr = np.asarray(nn_ind)
        if arr.ndim != 2:
            raise ValueError("nn_ind must be a 2D neighbour index matrix")
        if not np.issubdtype(arr.dtype, np.integer):
            # the reference masks `< 0` on the original dtype, then casts with astype(int); what is negative after the
            # cast (inf, nan, huge values) makes np.bincount raise
            neg = arr < 0
            with np.errstate(invalid="ignore"):
                cast = arr.astype(np.int64)
            if np.any((cast < 0) & ~neg):
                raise ValueError("'list' argument must have no negative elements")
            arr = cast
        ctx = ctx or N.Context.get()
        ind = ctx.to_device(np.ascontiguousarray(arr, dtype=np.int64))
    n_train, cols = ind.shape
    n_test = target_samples
    if k is None:
        k = cols
    elif k > cols:
        k = cols
        warnings.warn(f"k > nn_ind.shape[1], k will be set to {k}", stacklevel=2)
    lib = ctx.lib
    lo, hi = C.c_int64(0), C.c_int64(0)
    # the histogram is sized by the ids of the first k columns only: the reference slices nn_ind[:, :k] before
    # np.bincount(..., minlength=n_train) (estimation.py:276-292)
    N._check(lib.kz_minmax_i64_2d(ctx.handle, ind.ptr, n_train, cols, int(k), C.byref(lo), C.byref(hi)), "kz_minmax_i64_2d")
    n_bins = max(n_train, int(hi.value) + 1)
    kocc = ctx.empty((n_bins,), np.int64)
    N._check(lib.kz_k_occurrence(ctx.handle, ind.ptr, n_train, cols, int(k), n_bins, kocc.ptr), "kz_k_occurrence")

    want_gini = return_value in ("gini", "all")
    thr = hub_size * k
    st = (C.c_double * 10)()
    N._check(lib.kz_kocc_stats(ctx.handle, kocc.ptr, n_bins, float(thr), int(want_gini), st), "kz_kocc_stats")
    total, abs_dev, m2s, m3s, sqrt_sum, kmax, n_zero, hub_sum, n_hub, gini_num = (float(x) for x in st)
    n = float(n_bins)
    mean = total / n
    m2, m3 = m2s / n, m3s / n
    k_skewness = m3 / m2 ** 1.5 if m2 > 0 else float("nan")                 # scipy.stats.skew (biased)
    std1 = math.sqrt(m2s / (n - 1.0)) if n > 1 else float("nan")             # k_occurrence.std(ddof=1)
    k_skewness_truncnorm = _truncnorm_third_moment((0.0 - mean) / std1) if std1 and std1 > 0 else float("nan")
    gini_index = gini_num / (2.0 * n * total) if want_gini else float("nan")  # estimation.py:98-99
    robinhood_index = 0.5 * abs_dev / total                                   # :126-128
    atkinson_index = float(1.0 - 1.0 / mean * (sqrt_sum / n) ** 2)            # :147-150 with eps = 0.5
    antihub_occurrence = n_zero / n                                           # :168-170
    hub_occurrence = hub_sum / k / n_test                                     # :192-194
    groupie_ratio = kmax / n_test / k                                         # :326

    def _select(mode):
        out = ctx.empty((n_bins,), np.int64)
        cnt = C.c_int64(0)
        N._check(lib.kz_kocc_select(ctx.handle, kocc.ptr, n_bins, mode, float(thr), out.ptr, C.byref(cnt)), "kz_kocc_select")
        return out.numpy()[: cnt.value]

    measures = {
        "k_skewness": k_skewness,
        "k_skewness_truncnorm": k_skewness_truncnorm,
        "atkinson": atkinson_index,
        "gini": gini_index,
        "robinhood": robinhood_index,
        "antihubs": None,
        "antihub_occurrence": antihub_occurrence,
        "hubs": None,
        "hub_occurrence": hub_occurrence,
        "groupie_ratio": groupie_ratio,
    }
    if return_value in ("all", "all_but_gini", "antihubs"):
        measures["antihubs"] = _select(0)
    if return_value in ("all", "all_but_gini", "hubs"):
        measures["hubs"] = _select(1)
    if store_k_occurrence or return_value == "k_occurrence":
        measures["k_occurrence"] = kocc.numpy()
    if return_value == "all":
        return measures
    if return_value == "all_but_gini":
        del measures["gini"]
        return measures
    if return_value not in measures:
        raise KeyError(f"unknown hubness measure {return_value}; valid: {VALID_HUBNESS_MEASURES}")
    return measures[return_value]
