"""ctypes binding of libkiez_amd.so (the C ABI declared in include/kiez_amd.h).

The HIP library is the product path: if it is missing, was built for another architecture, or no MI355X
is visible, every entry point raises — there is no CPU fallback anywhere in this package.

Import-order note (DESIGN.md "Process model"): PyTorch-ROCm wheels bundle their own HIP runtime.  When a
process uses both torch (for torch.distributed / RCCL) and this library, torch must be imported FIRST so
that both resolve to one HIP runtime; `load()` therefore imports torch before dlopen when it is already
importable and KIEZ_AMD_WITH_TORCH=1 is set (bench.py and kiez_amd.distributed do this).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

_LIB_NAME = "libkiez_amd.so"
_lib = None
_lock = threading.Lock()

KZ_F32, KZ_F64 = 0, 1
KZ_EUCLIDEAN, KZ_SQEUCLIDEAN, KZ_COSINE, KZ_MANHATTAN, KZ_CHEBYSHEV, KZ_MINKOWSKI = 0, 1, 2, 3, 4, 5
METRIC_IDS = {"euclidean": KZ_EUCLIDEAN, "sqeuclidean": KZ_SQEUCLIDEAN, "cosine": KZ_COSINE, "manhattan": KZ_MANHATTAN,
              "chebyshev": KZ_CHEBYSHEV, "minkowski": KZ_MINKOWSKI}


def split_metric(metric: str):
    """'minkowski[3.0]' -> ('minkowski', 3.0); any other canonical metric name -> (name, None)."""
    if metric.startswith("minkowski[") and metric.endswith("]"):
        return "minkowski", float(metric[len("minkowski["):-1])
    return metric, None
MAX_FUSED_NEIGHBORS = 110   # neighbours per query ONE fused list keeps (list length 128 minus the certification margin): the limit of
                            # the shared sweep and of the single-source split; kz_knn itself takes 111 .. ~540 on its long-k route
                            # (lists over many index ranges) and anything up to MAX_NEIGHBORS on the exact float64 kernels
MAX_NEIGHBORS = 4096
MAX_HUBNESS_CANDIDATES = 4096  # n_candidates the device hubness kernels (transform, final sort) handle (KZ_MAX_CANDIDATES)
MERGE_MAX_ENTRIES = 8192       # entries per row kz_merge_topk merges (KZ_MERGE_MAX_ENTRIES)
ABI_VERSION = 7

_ERR_TYPES = {1: ValueError, 2: RuntimeError, 3: NotImplementedError, 4: MemoryError, 5: ValueError}


class KnnStats(C.Structure):
    _fields_ = [
        ("main_kernel_ms", C.c_double),
        ("finalize_ms", C.c_double),
        ("fallback_ms", C.c_double),
        ("n_fallback_rows", C.c_int64),
        ("list_len", C.c_int32),
        ("n_splits", C.c_int32),
        ("n_blocks", C.c_int32),
        ("first_pass", C.c_int32),
        ("n_escalated_rows", C.c_int64),
        ("max_err_ratio", C.c_double),
        ("dual", C.c_int32),
        ("n_first_pass_fail", C.c_int32),
        ("n_events", C.c_int64),
        ("n_overflow_rows", C.c_int64),
        ("n_logged_groups", C.c_int64),
        ("wide_lists", C.c_int32),
        ("n_spec_rows", C.c_int32),
        ("probe_ms", C.c_double),
        ("n_range_rows", C.c_int64),
        ("n_range_pairs", C.c_int64),
        ("n_range_group_rows", C.c_int64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


# every symbol include/kiez_amd.h declares: (name, restype, argtypes)
_P = C.c_void_p
_I64 = C.c_int64
SYMBOLS = [
    ("kz_abi_version", C.c_int, []),
    ("kz_last_error", C.c_char_p, []),
    ("kz_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("kz_ctx_create", C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    ("kz_ctx_destroy", C.c_int, [_P]),
    ("kz_ctx_sync", C.c_int, [_P]),
    ("kz_ctx_trim", C.c_int, [_P]),
    ("kz_ctx_set_option", C.c_int, [_P, C.c_char_p, C.c_double]),
    ("kz_malloc", C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    ("kz_free", C.c_int, [_P, _P]),
    ("kz_memcpy_h2d", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("kz_memcpy_d2h", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("kz_memcpy_d2d", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("kz_matrix_create", C.c_int, [_P, _P, C.c_int, _I64, _I64, C.c_int, C.c_int, C.POINTER(_P)]),
    ("kz_matrix_destroy", C.c_int, [_P]),
    ("kz_matrix_set_minkowski_p", C.c_int, [_P, C.c_double]),
    ("kz_matrix_shape", C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("kz_knn", C.c_int, [_P, _P, _I64, _I64, _P, C.c_int, C.c_int, _P, _P, C.POINTER(KnnStats)]),
    ("kz_knn_dual", C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, _P, C.POINTER(KnnStats), C.POINTER(KnnStats)]),
    ("kz_pair_values", C.c_int, [_P, _P, _I64, _I64, _P, _P, C.c_int, _P]),
    ("kz_merge_topk", C.c_int, [_P, _P, _P, _P, _I64, C.c_int, C.c_int, C.c_int, _P, _P]),
    ("kz_split_self", C.c_int, [_P, _P, _P, _I64, C.c_int, _I64, _P, _P, _P, _P]),
    ("kz_knn_plan", C.c_int, [_I64, _I64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                              C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("kz_selftest_div", C.c_int, [_P, _I64, C.c_uint64, C.c_int, C.POINTER(_I64)]),
    ("kz_row_stats", C.c_int, [_P, _P, _I64, C.c_int, _P, _P, _P]),
    ("kz_csls", C.c_int, [_P, _P, _P, _I64, C.c_int, _P, _P]),
    ("kz_local_scaling", C.c_int, [_P, _P, _P, _I64, C.c_int, _P, C.c_int, _P]),
    ("kz_mp_normal", C.c_int, [_P, _P, _P, _I64, C.c_int, _P, _P, _P]),
    ("kz_mp_empiric", C.c_int, [_P, _P, _P, _I64, C.c_int, _P, _P, _I64, C.c_int, _P]),
    ("kz_dsl_fit", C.c_int, [_P, _P, _I64, C.c_int, _P, _P, _I64, _P]),
    ("kz_dsl_transform", C.c_int, [_P, _P, _I64, C.c_int, _P, _I64, _P, _P, _P, _P]),
    ("kz_dsl_finalize", C.c_int, [_P, _P, _I64, C.c_double, C.c_int]),
    ("kz_select_topk", C.c_int, [_P, _P, _P, _I64, C.c_int, C.c_int, _P, _P]),
    ("kz_cast_f64_f32", C.c_int, [_P, _P, _P, _I64]),
    ("kz_minmax_i64", C.c_int, [_P, _P, _I64, C.POINTER(_I64), C.POINTER(_I64)]),
    ("kz_minmax_i64_2d", C.c_int, [_P, _P, _I64, C.c_int, C.c_int, C.POINTER(_I64), C.POINTER(_I64)]),
    ("kz_k_occurrence", C.c_int, [_P, _P, _I64, C.c_int, C.c_int, _I64, _P]),
    ("kz_kocc_stats", C.c_int, [_P, _P, _I64, C.c_double, C.c_int, C.POINTER(C.c_double)]),
    ("kz_kocc_select", C.c_int, [_P, _P, _I64, C.c_int, C.c_double, _P, C.POINTER(_I64)]),
    ("kz_hit_positions", C.c_int, [_P, _P, _P, _I64, C.c_int, _P]),
    ("kz_comm_unique_id", C.c_int, [_P]),
    ("kz_comm_create", C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    ("kz_comm_destroy", C.c_int, [_P]),
    ("kz_comm_rank", C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("kz_comm_broadcast", C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    ("kz_comm_all_gather", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("kz_comm_all_to_all", C.c_int, [_P, _P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _P, C.c_size_t]),
    ("kz_comm_all_reduce_min_f64", C.c_int, [_P, _P, C.c_size_t]),
]


def library_path() -> Path:
    override = os.environ.get("KIEZ_AMD_LIB")  # diagnostic builds (tools/ablate.sh)
    return Path(override) if override else Path(__file__).resolve().parent / _LIB_NAME


def load():
    """dlopen libkiez_amd.so and bind every symbol (no GPU call is made here)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not path.exists():
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C kiez_amd/csrc`).  kiez_amd has no CPU fallback.")
        if os.environ.get("KIEZ_AMD_WITH_TORCH") == "1" and "torch" not in sys.modules:
            import torch  # noqa: F401  (one HIP runtime per process: torch's must be loaded first)
        lib = C.CDLL(str(path))
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.kz_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version {lib.kz_abi_version()} != {ABI_VERSION}")
        _lib = lib
    return _lib


def _check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().kz_last_error().decode("utf-8", "replace")
        raise _ERR_TYPES.get(rc, RuntimeError)(msg or f"{what} failed with status {rc}")


class Context:
    """One GPU (kz_ctx).  Contexts are cached per (device, stream)."""

    _cache = {}

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        lib = load()
        n = C.c_int(0)
        rc = lib.kz_device_count(C.byref(n))
        if rc != 0 or n.value < 1:
            raise RuntimeError(
                "kiez_amd: no MI355X (gfx950) device is visible to HIP; the exact-kNN path runs on the GPU only "
                f"({lib.kz_last_error().decode('utf-8', 'replace')})")
        h = _P()
        _check(lib.kz_ctx_create(int(device), _P(stream) if stream else None, C.byref(h)), "kz_ctx_create")
        self.lib = lib
        self.handle = h
        self.device = int(device)

    @classmethod
    def get(cls, device: Optional[int] = None, stream: Optional[int] = None) -> "Context":
        if device is None:
            device = int(os.environ.get("KIEZ_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = C.c_int(0)
            if load().kz_device_count(C.byref(n)) == 0 and n.value > 0:
                device %= n.value
        key = (device, stream)
        ctx = cls._cache.get(key)
        if ctx is None:
            ctx = cls(device, stream)
            cls._cache[key] = ctx
        return ctx

    def set_option(self, name: str, value: float):
        _check(self.lib.kz_ctx_set_option(self.handle, name.encode(), float(value)), "kz_ctx_set_option")

    def sync(self):
        _check(self.lib.kz_ctx_sync(self.handle), "kz_ctx_sync")

    def trim(self):
        """Release the context's cached device buffers (kz_ctx_trim)."""
        _check(self.lib.kz_ctx_trim(self.handle), "kz_ctx_trim")

    # ---- device arrays -------------------------------------------------------------------------------
    def empty(self, shape, dtype) -> "DeviceArray":
        return DeviceArray(self, shape, dtype)

    def to_device(self, arr: np.ndarray) -> "DeviceArray":
        arr = np.ascontiguousarray(arr)
        out = DeviceArray(self, arr.shape, arr.dtype)
        if arr.nbytes:
            _check(self.lib.kz_memcpy_h2d(self.handle, out.ptr, arr.ctypes.data_as(_P), arr.nbytes), "kz_memcpy_h2d")
        return out

    def as_device(self, arr, dtype=None) -> "DeviceArray":
        """numpy array -> uploaded copy; DeviceArray -> itself."""
        if isinstance(arr, DeviceArray):
            if dtype is not None and arr.dtype != np.dtype(dtype):
                raise TypeError(f"device array has dtype {arr.dtype}, expected {np.dtype(dtype)}")
            return arr
        a = np.asarray(arr)
        if dtype is not None:
            a = a.astype(dtype, copy=False)
        return self.to_device(a)


class DeviceArray:
    """A dense array in HBM owned by this object (freed on garbage collection)."""

    def __init__(self, ctx: Context, shape, dtype, ptr: Optional[int] = None, owner=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._owner = owner
        if ptr is None:
            p = _P()
            _check(ctx.lib.kz_malloc(ctx.handle, max(self.nbytes, 16), C.byref(p)), "kz_malloc")
            self.ptr = p
            self._owned = True
        else:
            self.ptr = _P(ptr)
            self._owned = False

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr:
            try:
                self.ctx.lib.kz_free(self.ctx.handle, self.ptr)
            except Exception:
                pass
            self._owned = False

    @property
    def __cuda_array_interface__(self):
        """Zero-copy export (torch.as_tensor(device_array, device="cuda"), cupy, numba)."""
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr.value or 0, False), "version": 3,
                "strides": None}

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            _check(self.ctx.lib.kz_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(_P), self.ptr, self.nbytes), "kz_memcpy_d2h")
        return out

    def view_rows(self, begin: int, count: int) -> "DeviceArray":
        """A non-owning view of rows [begin, begin+count)."""
        row_bytes = self.nbytes // max(self.shape[0], 1)
        return DeviceArray(self.ctx, (count,) + self.shape[1:], self.dtype, ptr=(self.ptr.value or 0) + begin * row_bytes,
                           owner=self)

    def copy_from(self, other: "DeviceArray"):
        assert other.nbytes == self.nbytes
        _check(self.ctx.lib.kz_memcpy_d2d(self.ctx.handle, self.ptr, other.ptr, self.nbytes), "kz_memcpy_d2d")

    def fill_from_host(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes
        _check(self.ctx.lib.kz_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data_as(_P), arr.nbytes), "kz_memcpy_h2d")


class DeviceMatrix:
    """kz_matrix: an embedding matrix in HBM (exact rows + float64 norms + MFMA operand images: float32 and split-bf16)."""

    def __init__(self, ctx: Context, data, metric: str, device_ptr: Optional[int] = None, shape=None, dtype=None,
                 borrow: bool = False, keepalive=None, rows_only: bool = False):
        """`device_ptr` + `borrow=True`: zero-copy -- the matrix reads the caller's HBM buffer in place (kz_matrix_create
        rows_on_device = 2); `keepalive` (the tensor / array that owns it) is held until the matrix is destroyed and must
        not be modified meanwhile (the reference holds its inputs the same way, neighbor_algorithm_base.py:95-96)."""
        self.ctx = ctx
        self.metric = metric
        metric, mink_p = split_metric(metric)
        h = _P()
        if device_ptr is None:
            arr = np.ascontiguousarray(data)
            if arr.ndim != 2:
                raise ValueError(f"Expected 2D array, got {arr.ndim}D array instead")
            if arr.dtype not in (np.float32, np.float64):
                arr = arr.astype(np.float64)
            self.shape = arr.shape
            self.dtype = arr.dtype
            _check(ctx.lib.kz_matrix_create(ctx.handle, arr.ctypes.data_as(_P), 0, arr.shape[0], arr.shape[1],
                                            KZ_F32 if arr.dtype == np.float32 else KZ_F64, METRIC_IDS[metric], C.byref(h)),
                   "kz_matrix_create")
        else:
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self._keepalive = keepalive if borrow else None
            if rows_only and not borrow:
                raise ValueError("rows_only needs a borrowed device buffer")
            # (rows_only: kz_matrix_create rows_on_device = 3 -- a row source for kz_dsl_fit, nothing computed, not searchable)
            _check(ctx.lib.kz_matrix_create(ctx.handle, _P(device_ptr), (3 if rows_only else 2) if borrow else 1, self.shape[0], self.shape[1],
                                            KZ_F32 if self.dtype == np.float32 else KZ_F64, METRIC_IDS[metric], C.byref(h)),
                   "kz_matrix_create")
        self.handle = h
        if mink_p is not None:
            _check(ctx.lib.kz_matrix_set_minkowski_p(h, mink_p), "kz_matrix_set_minkowski_p")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                self.ctx.lib.kz_matrix_destroy(h)
            except Exception:
                pass
            self.handle = None


def knn(ctx: Context, query: DeviceMatrix, index: DeviceMatrix, k: int, exclude_self: bool = False,
        q_begin: int = 0, q_count: Optional[int] = None) -> Tuple[DeviceArray, DeviceArray, dict]:
    """kz_knn -> (dist float64 [q, k], ind int64 [q, k], stats) on the device."""
    if q_count is None:
        q_count = query.shape[0] - q_begin
    dist = ctx.empty((q_count, k), np.float64)
    ind = ctx.empty((q_count, k), np.int64)
    st = KnnStats()
    _check(ctx.lib.kz_knn(ctx.handle, query.handle, q_begin, q_count, index.handle, int(k), int(bool(exclude_self)),
                          dist.ptr, ind.ptr, C.byref(st)), "kz_knn")
    return dist, ind, st.as_dict()


def knn_dual(ctx: Context, a: DeviceMatrix, b: DeviceMatrix, k: int):
    """kz_knn_dual: both directions between two matrices from one sweep of the distance matrix.
    Returns ((dist, ind, stats) of a -> b [a.n, k], (dist, ind, stats) of b -> a [b.n, k])."""
    d_ab = ctx.empty((a.shape[0], k), np.float64)
    i_ab = ctx.empty((a.shape[0], k), np.int64)
    d_ba = ctx.empty((b.shape[0], k), np.float64)
    i_ba = ctx.empty((b.shape[0], k), np.int64)
    s_ab, s_ba = KnnStats(), KnnStats()
    _check(ctx.lib.kz_knn_dual(ctx.handle, a.handle, b.handle, int(k), d_ab.ptr, i_ab.ptr, d_ba.ptr, i_ba.ptr,
                               C.byref(s_ab), C.byref(s_ba)), "kz_knn_dual")
    return (d_ab, i_ab, s_ab.as_dict()), (d_ba, i_ba, s_ba.as_dict())


def split_self(ctx: Context, dist: DeviceArray, ind: DeviceArray, row0: int = 0):
    """kz_split_self: [n, K + 1] self-search without stripping -> ((rev_dist, rev_ind), (fwd_dist, fwd_ind)), each [n, K]."""
    n, k1 = dist.shape
    outs = [ctx.empty((n, k1 - 1), np.float64), ctx.empty((n, k1 - 1), np.int64), ctx.empty((n, k1 - 1), np.float64),
            ctx.empty((n, k1 - 1), np.int64)]
    _check(ctx.lib.kz_split_self(ctx.handle, dist.ptr, ind.ptr, n, int(k1), int(row0), outs[0].ptr, outs[1].ptr, outs[2].ptr,
                                 outs[3].ptr), "kz_split_self")
    return (outs[0], outs[1]), (outs[2], outs[3])


def row_stats(ctx: Context, dist: DeviceArray, mean=False, std=False, last=False):
    n, K = dist.shape
    m = ctx.empty((n,), np.float64) if mean else None
    s = ctx.empty((n,), np.float64) if std else None
    l_ = ctx.empty((n,), np.float64) if last else None
    _check(ctx.lib.kz_row_stats(ctx.handle, dist.ptr, n, K, m.ptr if m else None, s.ptr if s else None,
                                l_.ptr if l_ else None), "kz_row_stats")
    return m, s, l_


def select_topk(ctx: Context, dist: DeviceArray, ind: DeviceArray, k: int):
    n, K = dist.shape
    od = ctx.empty((n, k), np.float64)
    oi = ctx.empty((n, k), np.int64)
    _check(ctx.lib.kz_select_topk(ctx.handle, dist.ptr, ind.ptr, n, K, int(k), od.ptr, oi.ptr), "kz_select_topk")
    return od, oi


def cast_f32(ctx: Context, arr: DeviceArray) -> DeviceArray:
    out = ctx.empty(arr.shape, np.float32)
    _check(ctx.lib.kz_cast_f64_f32(ctx.handle, arr.ptr, out.ptr, int(np.prod(arr.shape, dtype=np.int64))), "kz_cast_f64_f32")
    return out
