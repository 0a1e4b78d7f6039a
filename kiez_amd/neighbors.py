"""NN backends: the reference's `NNAlgorithm` plugin interface and the MI355X exact backend behind it.

`NNAlgorithm` mirrors kiez/neighbors/neighbor_algorithm_base.py:13-136 (same method names, argument
meaning, errors and warnings).  `SklearnNN` keeps the reference class's name and constructor
(kiez/neighbors/exact/sklearn_nearest_neighbors.py:51-65) so that `Kiez(algorithm="SklearnNN", ...)`
configurations keep working, but `_fit` / `_kneighbors` run on the GPU through the C ABI
(include/kiez_amd.h): brute-force exact search, identical neighbours.
"""
from __future__ import annotations

import warnings
from abc import ABC, abstractmethod
from typing import Any, Optional, Tuple

import numpy as np

from . import _native as N


class NotFittedError(ValueError, AttributeError):
    """Same bases as sklearn.exceptions.NotFittedError (raised by the reference through check_is_fitted)."""


def check_is_fitted(obj, attributes, all_or_any=all):
    if isinstance(attributes, str):
        attributes = [attributes]
    if not all_or_any([hasattr(obj, a) for a in attributes]):
        raise NotFittedError(
            f"This {type(obj).__name__} instance is not fitted yet. Call 'fit' with appropriate arguments before "
            "using this estimator.")


class NNAlgorithm(ABC):
    """The reference's plugin interface for nearest-neighbour backends (kiez/neighbors/neighbor_algorithm_base.py:13-136).

    A backend implements `_fit(data, is_source) -> index` and `_kneighbors(k, query, index, return_distance,
    is_self_querying)`; this base class keeps the two fitted sides and does the argument checking.  Attributes after `fit`
    are the reference's: `source_`, `target_` (the caller's arrays), `source_index`, `target_index`,
    `source_equals_target`.  Organisation is this repository's own: one table of the two search directions
    (`_direction`) instead of branches, checks as small helpers.
    """

    _ALLOWED_INPUT_TYPES: Tuple[Any, ...] = (np.ndarray,)

    def __init__(self, n_candidates, metric, n_jobs):
        self.n_candidates = n_candidates
        self.metric = metric
        self.n_jobs = n_jobs

    # ---- what a backend provides --------------------------------------------------------------------
    @property
    @abstractmethod
    def valid_metrics(self):
        pass  # pragma: no cover

    @abstractmethod
    def _fit(self, data, is_source: bool) -> Any:
        pass  # pragma: no cover

    @abstractmethod
    def _kneighbors(self, k, query, index, return_distance, is_self_querying):
        pass  # pragma: no cover

    # ---- checks -------------------------------------------------------------------------------------
    def _check_input_types(self, value):
        values = value if isinstance(value, tuple) else (value,)
        allowed = type(self)._ALLOWED_INPUT_TYPES
        if any(x is not None and not isinstance(x, allowed) for x in values):
            found_types = [type(x) for x in values]
            raise ValueError(f"Not implemented for input type(s) {found_types}! Only {allowed} allowed!")

    @staticmethod
    def _same_width(source, target):
        if source.shape[1] != target.shape[1]:
            raise ValueError("Expected source and target to have the same number of features,"
                             f" but got source.shape: {source.shape} and target.shape: {target.shape}")

    def _check_k_value(self, k: int, needed_space) -> int:
        if not np.issubdtype(type(k), np.integer):
            raise TypeError(f"k does not take {type(k)} value, enter integer value")
        if k <= 0:
            raise ValueError(f"Expected k > 0. Got {k}")
        if k <= needed_space:
            return k
        warnings.warn(f"k={k} is larger than number of samples in indexed space.\n" + f"Setting to k={needed_space}",
                      stacklevel=2)
        return needed_space

    def _describe_source_target_fitted(self):
        if not hasattr(self, "source_"):
            return " is unfitted"
        return f" is fitted with: source.shape={self.source_.shape} and target.shape={self.target_.shape}"

    # ---- fit ----------------------------------------------------------------------------------------
    def fit(self, source, target=None, only_fit_target: bool = False):
        """Index the data.  One argument: a single-source search (the index serves both directions).  Two arguments: both
        sides are indexed, or only the target when the caller never searches target -> source (`only_fit_target`, used by
        NoHubnessReduction).  An index that this call does not rebuild is dropped, never left over from an earlier fit."""
        self._check_input_types((source, target))
        single = target is None
        if not single:
            self._same_width(source, target)
        for stale in ("source_index", "target_index"):
            if hasattr(self, stale):
                delattr(self, stale)
        if single:
            self.source_index = self.target_index = self._fit(source, True)
            target = source
        elif only_fit_target:
            self.target_index = self._fit(target, True)
        else:
            self.source_index = self._fit(source, True)
            self.target_index = self._fit(target, False)
        self.source_equals_target = single
        self.source_, self.target_ = source, target

    # ---- search -------------------------------------------------------------------------------------
    def _direction(self, s_to_t: bool):
        """(default query array, index searched, its size) of a search direction."""
        if s_to_t:
            return self.source_, getattr(self, "target_index", None), self.target_.shape[0]
        return self.target_, getattr(self, "source_index", None), self.source_.shape[0]

    def _select_direction(self, k, query, s_to_t):
        check_is_fitted(self, ["source_index", "target_index"], all_or_any=any)
        default_query, index, index_size = self._direction(s_to_t)
        if index is None:   # e.g. target -> source after fit(..., only_fit_target=True)
            raise NotFittedError(f"'{type(self).__name__}' object has no attribute "
                                 f"'{'target_index' if s_to_t else 'source_index'}'")
        # only the implicit query of a single-source fit searches "itself" (the explicit reverse pass of
        # HubnessReduction.fit keeps each row as its own first neighbour, base.py:37-42)
        is_self_querying = query is None and self.source_equals_target
        k = self._check_k_value(self.n_candidates if k is None else k, index_size)
        return k, (default_query if query is None else query), index, is_self_querying

    def kneighbors(self, k=None, query=None, s_to_t=True, return_distance=True):
        """k nearest neighbours of `query` (default: the fitted source, or target when `s_to_t` is False)."""
        k, query, index, is_self_querying = self._select_direction(k, query, s_to_t)
        return self._kneighbors(k=k, query=query, index=index, return_distance=return_distance,
                                is_self_querying=is_self_querying)


def _torch_if_loaded():
    """torch, but only if the caller already imported it (this package never imports torch on its own: the HIP runtime
    bundled with torch has to be loaded BEFORE libkiez_amd.so, kiez_amd/_native.py)."""
    import sys
    return sys.modules.get("torch")


def _is_tensor(x) -> bool:
    t = _torch_if_loaded()
    return t is not None and isinstance(x, t.Tensor)


def canonical_metric(metric: str, p=2) -> str:
    """Map the reference's metric spelling to one the HIP kernels implement; anything else fails loudly.

    scikit-learn's own aliases (sklearn/metrics/_dist_metrics.pyx.tp, DistanceMetric.get_metric): minkowski with p = 1 / 2 / inf IS
    manhattan / euclidean / chebyshev; `p` is ignored for every other metric name.  Euclidean, squared euclidean and cosine run the
    fused MFMA kernels; manhattan, chebyshev and minkowski(p) have no inner-product form and run on a register-tiled VALU kernel + the exact selection."""
    if metric == "minkowski":
        if not isinstance(p, (int, float, np.integer, np.floating)) or isinstance(p, bool) or not p >= 1:
            raise ValueError(f"metric='minkowski' needs p >= 1 on the MI355X exact backend (got p={p!r})")
        if p == 2:
            return "euclidean"
        if p == 1:
            return "manhattan"
        if np.isinf(p):
            return "chebyshev"
        if p >= 1e6:    # (kz_matrix_set_minkowski_p's limit: said here, in the constructor, like every other metric error)
            raise ValueError(f"metric='minkowski' with p={p!r}: the MI355X exact backend takes 1 <= p < 1e6 (or p = inf: chebyshev)")
        return f"minkowski[{float(p)!r}]"
    if metric in ("l2", "euclidean"):
        return "euclidean"
    if metric in ("manhattan", "cityblock", "l1"):
        return "manhattan"
    if metric in ("sqeuclidean", "cosine", "chebyshev"):
        return metric
    raise ValueError(
        f"metric='{metric}' is not implemented by the MI355X exact backend; valid metrics: {SklearnNN.valid_metrics}")


class SklearnNN(NNAlgorithm):
    """Exact brute-force nearest neighbours on MI355X, under the reference class's name and signature
    (kiez/neighbors/exact/sklearn_nearest_neighbors.py:7-101).

    `algorithm`, `leaf_size` and `n_jobs` are accepted for configuration compatibility; every sklearn
    algorithm choice is exact, so the neighbours are the same and the search always runs as one fused
    distance + top-k pass on the GPU.
    """

    valid_metrics = ["chebyshev", "cityblock", "cosine", "euclidean", "l1", "l2", "manhattan", "minkowski", "sqeuclidean"]
    # numpy arrays as in the reference; additionally arrays already resident in HBM (zero-copy fit)
    _ALLOWED_INPUT_TYPES = (np.ndarray, N.DeviceArray)

    def __init__(self, n_candidates=5, algorithm="auto", leaf_size=30, metric="minkowski", p=2, metric_params=None,
                 n_jobs=None, device=None):
        super().__init__(n_candidates=n_candidates, metric=metric, n_jobs=n_jobs)
        self.algorithm = algorithm
        self.leaf_size = leaf_size
        self.p = p
        self.metric_params = metric_params
        self.device = device
        self._metric_c = canonical_metric(metric, p)
        if metric_params:
            raise NotImplementedError(f"metric_params={metric_params!r} is not implemented by the MI355X exact backend")
        if isinstance(n_candidates, (int, np.integer)) and n_candidates > N.MAX_NEIGHBORS - 1:
            raise NotImplementedError(f"n_candidates={n_candidates} exceeds the {N.MAX_NEIGHBORS - 1} neighbours per query the "
                                      "MI355X exact backend supports")
        self._ctx = None
        self._aux = None  # (array, DeviceMatrix) of the most recent query array that is not a fitted side
        self._forward = None  # (k, dist, ind): source -> target result that came out of a shared sweep (kneighbors_device_both)
        self.last_stats = None
        self.last_stats_reverse = None   # statistics of the target -> source direction of the last shared sweep (else None)

    def __repr__(self):
        return (f"{self.__class__.__name__}(n_candidates={self.n_candidates},algorithm={self.algorithm},"
                f"leaf_size={self.leaf_size},metric={self.metric},n_jobs={self.n_jobs} )")

    # ---- device plumbing ---------------------------------------------------------------------------
    @property
    def ctx(self) -> N.Context:
        if self._ctx is None:
            self._ctx = N.Context.get(self.device)
        return self._ctx

    @staticmethod
    def _prepare(data) -> np.ndarray:
        arr = np.asarray(data)
        if arr.ndim != 2:
            raise ValueError(f"Expected 2D array, got {arr.ndim}D array instead")
        if arr.dtype not in (np.float32, np.float64):
            arr = arr.astype(np.float64)
        return np.ascontiguousarray(arr)

    def _check_input_types(self, value):
        # torch tensors are accepted like the reference's Faiss backend does (kiez/neighbors/approximate/faiss.py:64-65):
        # CUDA tensors are consumed in place (zero-copy fit), CPU tensors through their numpy view
        if not isinstance(value, tuple):
            value = (value,)
        super()._check_input_types(tuple(None if _is_tensor(x) else x for x in value))

    def _make_matrix(self, data, dtype=None) -> N.DeviceMatrix:
        if _is_tensor(data):
            torch = _torch_if_loaded()
            t = data.detach()
            if t.dim() != 2:
                raise ValueError(f"Expected 2D array, got {t.dim()}D array instead")
            if t.dtype not in (torch.float32, torch.float64):
                t = t.to(torch.float64)
            if dtype is not None and np.dtype(str(t.dtype).replace("torch.", "")) != dtype:
                t = t.to(torch.float32 if dtype == np.float32 else torch.float64)
            if not t.is_cuda:
                return self._make_matrix(t.contiguous().numpy(), dtype)
            t = t.contiguous()
            if t.device.index != self.ctx.device:
                raise ValueError(f"tensor lives on cuda:{t.device.index}, the NN backend on device {self.ctx.device}")
            torch.cuda.current_stream(t.device).synchronize()   # the producer stream is not ours
            # zero-copy: the matrix reads the tensor's HBM in place and keeps it alive (the reference, too, only keeps
            # references to its inputs, neighbor_algorithm_base.py:95-96: they must not be modified while fitted)
            return N.DeviceMatrix(self.ctx, None, self._metric_c, device_ptr=t.data_ptr(), shape=tuple(t.shape),
                                  dtype=np.float32 if t.dtype == torch.float32 else np.float64, borrow=True, keepalive=t)
        if isinstance(data, N.DeviceArray):
            if len(data.shape) != 2 or data.dtype not in (np.float32, np.float64):
                raise ValueError("device inputs must be 2D float32/float64 arrays")
            if dtype is not None and data.dtype != dtype:
                raise ValueError(f"device input has dtype {data.dtype}, the index has {dtype}")
            return N.DeviceMatrix(self.ctx, None, self._metric_c, device_ptr=data.ptr.value, shape=data.shape,
                                  dtype=data.dtype, borrow=True, keepalive=data)
        arr = self._prepare(data)
        if dtype is not None and arr.dtype != dtype:
            arr = arr.astype(dtype)
        return N.DeviceMatrix(self.ctx, arr, self._metric_c)

    def _fit(self, data, is_source: bool):
        """Replaces SklearnNN._fit (sklearn_nearest_neighbors.py:83-94): upload + norms + MFMA tile packing."""
        self._aux = None
        self._forward = None
        self.last_stats_reverse = None
        return self._make_matrix(data)

    def fit(self, source, target=None, only_fit_target: bool = False):
        self._check_input_types((source, target))
        if (isinstance(source, np.ndarray) and isinstance(target, np.ndarray) and source.ndim == 2 and target.ndim == 2
                and source.shape[1] == target.shape[1] and source.dtype != target.dtype):
            # one dtype for both sides (sklearn dispatches on X.dtype == Y.dtype, _dispatcher.py:292): use float64
            super().fit(np.asarray(source, dtype=np.float64), np.asarray(target, dtype=np.float64), only_fit_target)
            self.source_, self.target_ = source, target  # keep the caller's arrays, as the reference does
            return
        super().fit(source, target, only_fit_target)

    def _matrix_for(self, array) -> N.DeviceMatrix:
        """Device matrix of a query array: the index of a fitted side if it IS that side's array, else an upload.  The most
        recent ad-hoc query is kept (a hubness transform asks for the same array right after the search) and replaced by
        the next one: nothing accumulates, and a new fit drops it."""
        if hasattr(self, "source_index") and array is self.source_:
            return self.source_index
        if hasattr(self, "target_index") and array is self.target_:
            return self.target_index
        if self._aux is not None and self._aux[0] is array:
            return self._aux[1]
        self._check_input_types(array)
        ref = self.target_index if hasattr(self, "target_index") else self.source_index
        m = self._make_matrix(array, dtype=ref.dtype)
        self._aux = (array, m)
        return m

    def _out_dtype(self, index: N.DeviceMatrix):
        # sklearn: euclidean family -> float64 always (ArgKmin); cosine -> dtype of the input
        return np.float32 if (self._metric_c == "cosine" and index.dtype == np.float32) else np.float64

    def kneighbors_device(self, k=None, query=None, s_to_t=True, q_begin=0, q_count=None):
        """`kneighbors` that leaves (dist float64, ind int64) in HBM; used by the GPU hubness reductions."""
        default_query = query is None
        k, query, index, is_self_querying = self._select_direction(k, query, s_to_t)
        if default_query and s_to_t and self._forward is not None:
            # the forward result of the shared sweep (kneighbors_device_both): handed out once, then dropped -- also when this
            # call asks for something else (another k, a row range): the cached [n_s, k] arrays must not stay pinned in HBM
            fk, dist, ind = self._forward
            self._forward = None
            if fk == k and q_begin == 0 and q_count is None:
                return dist, ind
            del dist, ind
        qm = self._matrix_for(query)
        dist, ind, stats = N.knn(self.ctx, qm, index, k, exclude_self=is_self_querying, q_begin=q_begin, q_count=q_count)
        self.last_stats = stats
        return dist, ind

    def kneighbors_device_both(self, k=None):
        """Both directions between the fitted source and target from ONE sweep of the distance matrix (kz_knn_dual): returns
        the target -> source result (what HubnessReduction.fit needs, base.py:37-42) and keeps the source -> target result
        for the `kneighbors_device(query=None, k)` call that follows (base.py:95-96).  Identical results to the two separate
        searches.  None when sharing does not apply (sides clamped to different k, more neighbours than the fused kernels keep)."""
        check_is_fitted(self, ["source_index", "target_index"], all_or_any=all)
        k = self.n_candidates if k is None else k
        if self.source_equals_target:
            # single source: the reverse search (explicit query: every row keeps itself) and the forward search (the row
            # itself stripped) are two views of ONE search for k + 1 neighbours (kz_split_self)
            n = self.source_.shape[0]
            if not np.issubdtype(type(k), np.integer) or k <= 0 or k + 1 > n or k + 1 > N.MAX_FUSED_NEIGHBORS:
                return None
            dist, ind, stats = N.knn(self.ctx, self.source_index, self.source_index, k + 1, exclude_self=False)
            self.last_stats = stats
            self.last_stats_reverse = None   # (one search serves both views: there is no second set of statistics)
            rev, fwd = N.split_self(self.ctx, dist, ind)
            self._forward = (k,) + fwd
            return rev
        n_s, n_t = self.source_.shape[0], self.target_.shape[0]
        if not np.issubdtype(type(k), np.integer) or k <= 0 or k > min(n_s, n_t):
            return None   # (the per-direction checks and clamps of kneighbors() apply: leave it to the separate searches)
        # the larger side sweeps as the query side: fewer rows get event buffers
        s_is_a = n_s >= n_t
        a, b = (self.source_index, self.target_index) if s_is_a else (self.target_index, self.source_index)
        (d_ab, i_ab, st_ab), (d_ba, i_ba, st_ba) = N.knn_dual(self.ctx, a, b, k)
        self.last_stats = st_ab
        self.last_stats_reverse = st_ba
        fwd, rev = ((d_ab, i_ab), (d_ba, i_ba)) if s_is_a else ((d_ba, i_ba), (d_ab, i_ab))
        self._forward = (k,) + fwd
        return rev

    def _kneighbors(self, k, query, index, return_distance, is_self_querying):
        """Replaces SklearnNN._kneighbors (sklearn_nearest_neighbors.py:96-101)."""
        qm = self._matrix_for(query)
        dist, ind, stats = N.knn(self.ctx, qm, index, k, exclude_self=is_self_querying)
        self.last_stats = stats
        if _is_tensor(query):   # tensors in -> tensors out (on the query's device), as the reference does with Faiss
            torch = _torch_if_loaded()
            self.ctx.sync()
            dev = query.device
            ind_t = torch.as_tensor(ind, device="cuda").clone().to(dev)
            dist_t = None
            if return_distance:
                dist_t = torch.as_tensor(dist, device="cuda").clone().to(dev)
                if self._out_dtype(index) == np.float32:
                    dist_t = dist_t.to(torch.float32)
            # the clones run on torch's stream, the pool that takes `dist` / `ind` back is ordered on OURS: finish them
            # before the DeviceArrays are released
            torch.cuda.current_stream().synchronize()
            return (dist_t, ind_t) if return_distance else ind_t
        ind_h = ind.numpy()
        if not return_distance:
            return ind_h
        dist_h = dist.numpy()
        out_dtype = self._out_dtype(index)
        if out_dtype != np.float64:
            dist_h = dist_h.astype(out_dtype)
        return dist_h, ind_h


def available_nn_algorithms(as_string: bool = False):
    """kiez/neighbors/util.py:18-39 (only the exact backend exists here; names are lower-cased as in the reference)."""
    return ["sklearnnn"] if as_string else [SklearnNN]
