"""Hubness reduction: the reference's `HubnessReduction` plugin interface with GPU implementations.

Mirrors kiez/hubness_reduction/{base,csls,mutual_proximity,local_scaling,dis_sim}.py: same class names,
constructor arguments, fitted attributes, errors and warnings.  `_fit` / `transform` accept what the
reference's do (numpy arrays from any `NNAlgorithm`) and additionally device arrays; when the NN backend is
the MI355X `SklearnNN`, `fit` / `kneighbors` keep every intermediate in HBM and only the final [n, k]
result crosses PCIe.
"""
from __future__ import annotations

import ctypes as C
import warnings
from abc import ABC, abstractmethod
from typing import Optional, Tuple

import numpy as np

from . import _native as N
from .neighbors import NNAlgorithm, SklearnNN, check_is_fitted

_P = C.c_void_p


def _is_dev(x):
    return isinstance(x, N.DeviceArray)


class HubnessReduction(ABC):
    """Base class for hubness reduction (kiez/hubness_reduction/base.py:17-105)."""

    def __init__(self, nn_algo: NNAlgorithm, verbose: int = 0, **kwargs):
        self.nn_algo = nn_algo
        self.verbose = verbose
        self._use_torch = False
        if nn_algo.n_candidates == 1:
            raise ValueError("Cannot perform hubness reduction with a single candidate per query!")
        # known at construction, so say it at construction (not after a long fit): the device transforms and the final
        # sort take up to KZ_MAX_CANDIDATES = 4096 candidates per query (the NN backend itself stops at 4095 neighbours)
        if (self._device_native and type(self).__name__ != "NoHubnessReduction" and isinstance(nn_algo.n_candidates, (int, np.integer))
                and nn_algo.n_candidates > N.MAX_HUBNESS_CANDIDATES):
            raise NotImplementedError(f"n_candidates={nn_algo.n_candidates}: the MI355X hubness reductions support up to "
                                      f"{N.MAX_HUBNESS_CANDIDATES} candidates per query")

    # Subclasses shipped here consume device arrays; a user-written subclass (docs/source/using_your_own.rst:11-19)
    # gets numpy arrays exactly as in the reference.
    _device_native = False
    # fit() takes its reverse search and the forward search of the kneighbors() that follows out of ONE sweep of the
    # distance matrix where the native library can (kz_knn_dual; results identical to the two searches).  Set to False on
    # an instance to search twice, as the reference does.
    _shared_sweep = True

    # ---- device helpers ---------------------------------------------------------------------------
    @property
    def _gpu_nn(self) -> bool:
        return self._device_native and isinstance(self.nn_algo, SklearnNN)

    @property
    def ctx(self) -> N.Context:
        if self._gpu_nn:
            return self.nn_algo.ctx
        if getattr(self, "_ctx", None) is None:
            self._ctx = N.Context.get()
        return self._ctx

    def _out_dtype(self, like):
        """dtype of the distances the reference would return for this input."""
        if _is_dev(like):
            return np.float64
        return np.float32 if np.asarray(like).dtype == np.float32 else np.float64

    @abstractmethod
    def _fit(self, neigh_dist, neigh_ind, source, target):
        pass  # pragma: no cover

    def fit(self, source, target=None):
        """base.py:33-50: index both sides, run the reverse (target -> source) kNN, hand it to `_fit`."""
        self.nn_algo.fit(source, target)
        if target is None:
            target = source
        if self._gpu_nn:
            # one sweep of the distance matrix serves this reverse search AND the forward search of kneighbors()
            # (kz_knn_dual); where that does not apply, the reverse search alone
            both = self.nn_algo.kneighbors_device_both(k=self.nn_algo.n_candidates) if self._shared_sweep else None
            if both is not None:
                neigh_dist_t_to_s, neigh_ind_t_to_s = both
            else:
                neigh_dist_t_to_s, neigh_ind_t_to_s = self.nn_algo.kneighbors_device(
                    k=self.nn_algo.n_candidates, query=target, s_to_t=False)
        else:
            neigh_dist_t_to_s, neigh_ind_t_to_s = self.nn_algo.kneighbors(
                k=self.nn_algo.n_candidates, query=target, s_to_t=False, return_distance=True)
        self._fit(neigh_dist_t_to_s, neigh_ind_t_to_s, source, target)

    @abstractmethod
    def transform(self, neigh_dist, neigh_ind, query) -> Tuple:
        pass  # pragma: no cover

    def _set_k_if_needed(self, k: Optional[int] = None) -> int:
        if k is None:
            warnings.warn(f"No k supplied, setting to n_candidates = {self.nn_algo.n_candidates}", stacklevel=2)
            return self.nn_algo.n_candidates
        if k > self.nn_algo.n_candidates:
            warnings.warn(f"k > n_candidates supplied! Setting to n_candidates = {self.nn_algo.n_candidates}", stacklevel=2)
            return self.nn_algo.n_candidates
        return k

    @staticmethod
    def _sort(hubness_reduced_query_dist, query_ind, n_neighbors: int, ctx: Optional[N.Context] = None):
        """base.py:72-87 (numpy branch) on the GPU: kz_select_topk.  numpy in -> numpy out; device in -> device out."""
        dev_in = _is_dev(hubness_reduced_query_dist)
        if ctx is None:
            ctx = hubness_reduced_query_dist.ctx if dev_in else N.Context.get()
        out_dtype = np.float64 if dev_in else np.asarray(hubness_reduced_query_dist).dtype
        d = ctx.as_device(hubness_reduced_query_dist, np.float64) if not dev_in else hubness_reduced_query_dist
        i = ctx.as_device(query_ind, np.int64)
        n_neighbors = min(int(n_neighbors), d.shape[1])
        od, oi = N.select_topk(ctx, d, i, n_neighbors)
        if dev_in:
            return od, oi
        od_h = od.numpy()
        if out_dtype == np.float32:
            od_h = od_h.astype(np.float32)
        return od_h, oi.numpy()

    def kneighbors_device(self, k: Optional[int] = None):
        """`kneighbors` with the result left in HBM (MI355X NN backend only): (dist, ind) DeviceArrays."""
        if not self._gpu_nn:
            raise TypeError("kneighbors_device needs the MI355X SklearnNN backend")
        n_neighbors = self._set_k_if_needed(k)
        nn = self.nn_algo
        query_dist, query_ind = nn.kneighbors_device(query=None, k=nn.n_candidates)
        hub_dist, query_ind = self.transform(query_dist, query_ind, nn.source_)
        od, oi = HubnessReduction._sort(hub_dist, query_ind, n_neighbors, ctx=self.ctx)
        if nn._out_dtype(nn.target_index) == np.float32:
            od = N.cast_f32(self.ctx, od)
        return od, oi

    def kneighbors(self, k: Optional[int] = None):
        """base.py:89-105: forward candidates, rescale, final top-k."""
        if self._gpu_nn:
            od, oi = self.kneighbors_device(k)
            src = self.nn_algo.source_
            from .neighbors import _is_tensor, _torch_if_loaded
            if _is_tensor(src):   # tensors in -> tensors out, on the source's device
                torch = _torch_if_loaded()
                self.ctx.sync()
                out = (torch.as_tensor(od, device="cuda").clone().to(src.device),
                       torch.as_tensor(oi, device="cuda").clone().to(src.device))
                torch.cuda.current_stream().synchronize()   # before od / oi go back to our stream-ordered pool
                return out
            return od.numpy(), oi.numpy()
        n_neighbors = self._set_k_if_needed(k)
        query_dist, query_ind = self.nn_algo.kneighbors(query=None, k=self.nn_algo.n_candidates, return_distance=True)
        hub_dist, query_ind = self.transform(query_dist, query_ind, self.nn_algo.source_)
        return HubnessReduction._sort(hub_dist, query_ind, n_neighbors, ctx=self.ctx)

    # ---- shared plumbing for the transform kernels -------------------------------------------------
    def _device_inputs(self, neigh_dist, neigh_ind):
        ctx = self.ctx
        return ctx.as_device(neigh_dist, np.float64), ctx.as_device(neigh_ind, np.int64)

    def _finish(self, out: N.DeviceArray, neigh_dist, neigh_ind):
        """device in -> device out; numpy in -> numpy out in the reference's dtype."""
        if _is_dev(neigh_dist):
            return out, neigh_ind
        res = out.numpy()
        dt = self._out_dtype(neigh_dist)
        if dt != np.float64:
            res = res.astype(dt)
        return res, neigh_ind


class NoHubnessReduction(HubnessReduction):
    """kiez/hubness_reduction/base.py:108-122: no reverse pass, the NN result is returned directly."""

    _device_native = True

    def _fit(self, neigh_dist, neigh_ind, source, target):
        pass  # pragma: no cover

    def fit(self, source, target=None):
        self.nn_algo.fit(source, target, only_fit_target=True)

    def transform(self, neigh_dist, neigh_ind, query):
        return neigh_dist, neigh_ind

    def kneighbors_device(self, k: Optional[int] = None):
        n_neighbors = self._set_k_if_needed(k)
        nn = self.nn_algo
        od, oi = nn.kneighbors_device(query=None, k=n_neighbors)
        if nn._out_dtype(nn.target_index) == np.float32:
            od = N.cast_f32(self.ctx, od)
        return od, oi

    def kneighbors(self, k: Optional[int] = None):
        n_neighbors = self._set_k_if_needed(k)
        return self.nn_algo.kneighbors(query=None, k=n_neighbors, return_distance=True)

    def __repr__(self):
        return f"{self.__class__.__name__}()"


class CSLS(HubnessReduction):
    """Cross-domain similarity local scaling (kiez/hubness_reduction/csls.py)."""

    _device_native = True

    def __repr__(self):
        return f"{self.__class__.__name__}(verbose = {self.verbose})"

    def _fit(self, neigh_dist, neigh_ind, source=None, target=None) -> "CSLS":
        self.r_dist_train_ = neigh_dist            # csls.py:53
        self.r_ind_train_ = neigh_ind              # csls.py:54
        d = self.ctx.as_device(neigh_dist, np.float64)
        self._r_train_dev, _, _ = N.row_stats(self.ctx, d, mean=True)   # csls.py:90, hoisted into fit
        return self

    def transform(self, neigh_dist, neigh_ind, query):
        check_is_fitted(self, "r_dist_train_")
        d, i = self._device_inputs(neigh_dist, neigh_ind)
        n, K = d.shape
        out = self.ctx.empty((n, K), np.float64)
        N._check(self.ctx.lib.kz_csls(self.ctx.handle, d.ptr, i.ptr, n, K, self._r_train_dev.ptr, out.ptr), "kz_csls")
        return self._finish(out, neigh_dist, neigh_ind)


class LocalScaling(HubnessReduction):
    """Local scaling / NICDM (kiez/hubness_reduction/local_scaling.py)."""

    _device_native = True

    def __init__(self, method: str = "standard", **kwargs):
        super().__init__(**kwargs)
        self.method = method.lower()
        if self.method not in ["ls", "standard", "nicdm"]:
            raise ValueError(f"Internal: Invalid method {self.method}. Try 'ls' or 'nicdm'.")

    def __repr__(self):
        return f"{self.__class__.__name__}(method = {self.method}, verbose = {self.verbose})"

    def _fit(self, neigh_dist, neigh_ind, source, target) -> "LocalScaling":
        self.r_dist_t_to_s_ = neigh_dist           # local_scaling.py:82
        self.r_ind_t_to_s_ = neigh_ind             # local_scaling.py:83
        d = self.ctx.as_device(neigh_dist, np.float64)
        if self.method == "nicdm":
            self._r_t_dev, _, _ = N.row_stats(self.ctx, d, mean=True)       # :143
        else:
            _, _, self._r_t_dev = N.row_stats(self.ctx, d, last=True)       # :136
        return self

    def transform(self, neigh_dist, neigh_ind, query=None):
        check_is_fitted(self, "r_dist_t_to_s_")
        d, i = self._device_inputs(neigh_dist, neigh_ind)
        n, K = d.shape
        out = self.ctx.empty((n, K), np.float64)
        N._check(self.ctx.lib.kz_local_scaling(self.ctx.handle, d.ptr, i.ptr, n, K, self._r_t_dev.ptr,
                                               1 if self.method == "nicdm" else 0, out.ptr), "kz_local_scaling")
        return self._finish(out, neigh_dist, neigh_ind)


class MutualProximity(HubnessReduction):
    """Mutual proximity, 'normal' and 'empiric' (kiez/hubness_reduction/mutual_proximity.py)."""

    _device_native = True

    def __init__(self, method: str = "normal", **kwargs):
        super().__init__(**kwargs)
        if method not in ["exact", "empiric", "normal", "gaussi"]:
            raise ValueError(f'Mutual proximity method "{method}" not recognized. Try "normal" or "empiric".')
        if method in ["exact", "empiric"]:
            self.method = "empiric"
        elif method in ["normal", "gaussi"]:
            self.method = "normal"

    def __repr__(self):
        return f"{self.__class__.__name__}(method = {self.method}, verbose = {self.verbose})"

    def _fit(self, neigh_dist, neigh_ind, source, target) -> "MutualProximity":
        self.n_train = neigh_dist.shape[0]
        d = self.ctx.as_device(neigh_dist, np.float64)
        if self.method == "empiric":
            self.neigh_dist_t_to_s_ = neigh_dist       # mutual_proximity.py:95
            self.neigh_ind_t_to_s_ = neigh_ind         # :96
            self._dist_t2s_dev = d
            self._ind_t2s_dev = self.ctx.as_device(neigh_ind, np.int64)
        else:
            mu, sd, _ = N.row_stats(self.ctx, d, mean=True, std=True)      # :102-103
            self._mu_dev, self._sd_dev = mu, sd
            self.mu_t_to_s_ = mu
            self.sd_t_to_s_ = sd
        return self

    def transform(self, neigh_dist, neigh_ind, query):
        check_is_fitted(self, ["mu_t_to_s_", "sd_t_to_s_", "neigh_dist_t_to_s_", "neigh_ind_t_to_s_"], all_or_any=any)
        d, i = self._device_inputs(neigh_dist, neigh_ind)
        n, K = d.shape
        out = self.ctx.empty((n, K), np.float64)
        if self.method == "normal":
            N._check(self.ctx.lib.kz_mp_normal(self.ctx.handle, d.ptr, i.ptr, n, K, self._mu_dev.ptr, self._sd_dev.ptr,
                                               out.ptr), "kz_mp_normal")
        else:
            n_t, Kt = self._dist_t2s_dev.shape
            N._check(self.ctx.lib.kz_mp_empiric(self.ctx.handle, d.ptr, i.ptr, n, K, self._dist_t2s_dev.ptr,
                                                self._ind_t2s_dev.ptr, n_t, Kt, out.ptr), "kz_mp_empiric")
        return self._finish(out, neigh_dist, neigh_ind)


_DESIRED_P_VALUE = 2


class DisSimLocal(HubnessReduction):
    """DisSimLocal (kiez/hubness_reduction/dis_sim.py)."""

    _device_native = True

    def __init__(self, squared: bool = True, **kwargs):
        super().__init__(**kwargs)
        self.squared = squared
        if self.nn_algo.metric in ["euclidean", "minkowski"]:
            self.squared = False
            if hasattr(self.nn_algo, "p") and self.nn_algo.p != _DESIRED_P_VALUE:
                raise ValueError(
                    "DisSimLocal only supports squared Euclidean distances. If the provided NNAlgorithm has a `p` "
                    f"parameter it must be set to p=2. Now it is p={self.nn_algo.p}")
        elif self.nn_algo.metric in ["sqeuclidean"]:
            self.squared = True
        else:
            raise ValueError(f"DisSimLocal only supports squared Euclidean distances, not metric={self.nn_algo.metric}.")

    def __repr__(self):
        return f"{self.__class__.__name__}(squared = {self.squared})"

    def _matrices(self, source, target):
        """Device matrices of the embeddings (reused from the NN backend when it is the GPU one)."""
        nn = self.nn_algo
        if self._gpu_nn and hasattr(nn, "source_index"):
            return nn._matrix_for(source), nn._matrix_for(target)
        s = np.asarray(source)
        t = np.asarray(target)
        if s.dtype != t.dtype or s.dtype not in (np.float32, np.float64):
            s, t = s.astype(np.float64), t.astype(np.float64)
        sm = N.DeviceMatrix(self.ctx, s, "sqeuclidean")
        tm = sm if target is source else N.DeviceMatrix(self.ctx, t, "sqeuclidean")
        return sm, tm

    def _fit(self, neigh_dist, neigh_ind, source, target) -> "DisSimLocal":
        ctx = self.ctx
        sm, tm = self._matrices(source, target)
        ind = ctx.as_device(neigh_ind, np.int64)
        n_t, Kt = ind.shape
        t2c = ctx.empty((n_t,), np.float64)
        N._check(ctx.lib.kz_dsl_fit(ctx.handle, ind.ptr, n_t, Kt, sm.handle, tm.handle, 0, t2c.ptr), "kz_dsl_fit")
        self.source_ = source
        self.target_ = target
        self._source_m, self._target_m = sm, tm
        self.target_dist_to_centroids_ = t2c       # dis_sim.py:107 (device array; .numpy() gives the reference's values)
        self.target_centroids_ = None              # not needed by transform; the reference only stores it
        return self

    def transform(self, neigh_dist, neigh_ind, query):
        check_is_fitted(self, ["target_", "target_dist_to_centroids_"])
        ctx = self.ctx
        i = ctx.as_device(neigh_ind, np.int64)
        n, K = i.shape
        if query is self.source_:
            qm = self._source_m
        elif self._gpu_nn:
            qm = self.nn_algo._matrix_for(query)
        else:
            qm = N.DeviceMatrix(ctx, np.asarray(query, dtype=self._target_m.dtype), "sqeuclidean")
        out = ctx.empty((n, K), np.float64)
        gmin = ctx.to_device(np.array([np.inf], dtype=np.float64))
        N._check(ctx.lib.kz_dsl_transform(ctx.handle, i.ptr, n, K, qm.handle, 0, self._target_m.handle,
                                          self.target_dist_to_centroids_.ptr, out.ptr, gmin.ptr), "kz_dsl_transform")
        mn = float(self._reduce_min(gmin.numpy()[0]))
        N._check(ctx.lib.kz_dsl_finalize(ctx.handle, out.ptr, n * K, mn, 1 if self.squared else 0), "kz_dsl_finalize")
        return self._finish(out, neigh_dist, neigh_ind)

    def _reduce_min(self, local_min: float) -> float:
        """Hook for the multi-GPU path: the shift uses the GLOBAL minimum (dis_sim.py:171-173)."""
        return local_min
