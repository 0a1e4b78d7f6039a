"""hits@k — `kiez.evaluate.hits` (kiez/evaluate/eval_metrics.py:23-61) with the row scan on the GPU.

For array input the neighbour matrix may live on the device (`Kiez.kneighbors_device`).  Dict input with arbitrary
labels is first encoded to integer ids on the host (data preparation); the matching itself runs in `kz_hit_positions`."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Union

import numpy as np

from . import _native as N

_NO_GOLD = np.iinfo(np.int64).min


def _as_row_number(x):
    """x as an integer if the reference's `==` would match it against an integer row number / neighbour id (ints, and floats with an
    integral value: 4.0 == 4 and hash(4.0) == hash(4), so `i in gold` and `gold[i] in nn_ind[i][:k]` both accept them), else None.
    Values that no int64 row number or neighbour id can equal -- beyond [-2**63, 2**63), e.g. 1e20 -- are None as well (the
    reference's `in` simply finds no hit for them; such pairs count in the denominator only).  bools stay what they are to
    Python's `==` and `hash`: True is 1 and False is 0 -- `1 in {True: t}` holds in the reference, too."""
    if isinstance(x, (bool, np.bool_)):
        return int(bool(x))
    if isinstance(x, (int, np.integer)):
        v = int(x)
    elif isinstance(x, (float, np.floating)) and np.isfinite(x) and float(x) == int(x):
        v = int(x)
    else:
        return None
    return v if -(1 << 63) <= v < (1 << 63) else None


def _gold_vector(gold: Dict[Any, Any], n_rows: int) -> np.ndarray:
    """gold target per source row (or _NO_GOLD): one vectorised scatter instead of a Python loop over all rows."""
    out = np.full(n_rows, _NO_GOLD, dtype=np.int64)
    # (array input: the reference tests `i in gold and gold[i] in nn_ind[i][:k]` with integer row numbers i, so only keys and
    #  targets that compare equal to an integer can ever hit -- ints and integral floats, e.g. gold read with np.loadtxt or from
    #  a pandas column; everything still counts in the denominator len(gold))
    pairs = [(a, b) for a, b in ((_as_row_number(k_), _as_row_number(v)) for k_, v in gold.items()) if a is not None and b is not None]
    if pairs:
        keys = np.fromiter((p[0] for p in pairs), dtype=np.int64, count=len(pairs))
        vals = np.fromiter((p[1] for p in pairs), dtype=np.int64, count=len(pairs))
        ok = (keys >= 0) & (keys < n_rows)
        out[keys[ok]] = vals[ok]
    return out


def hits(nn_ind: Union[np.ndarray, list, Dict[Any, List], "N.DeviceArray"], gold: Dict[Any, Any], k=None,
         ctx: Optional[N.Context] = None) -> Dict[int, float]:
    """Relative hits@k for every k in `k` (default [1, 5, 10]); same semantics as the reference."""
    if k is None:
        k = [1, 5, 10]
    k = sorted(k)
    if isinstance(nn_ind, dict):
        # encode labels: neighbour labels and gold targets share one code table; rows follow the dict order
        codes: Dict[Any, int] = {}

        def code(x):
            return codes.setdefault(x, len(codes))
        keys = list(nn_ind.keys())
        width = max((len(v) for v in nn_ind.values()), default=0)
        mat = np.full((len(keys), max(width, 1)), -1, dtype=np.int64)
        for r, key in enumerate(keys):
            row = [code(x) for x in nn_ind[key]]
            mat[r, : len(row)] = row
        gold_arr = np.array([code(gold[key]) if key in gold else _NO_GOLD for key in keys], dtype=np.int64)
        ind_host = mat
    elif isinstance(nn_ind, N.DeviceArray):
        ind_host = None
        n_rows = nn_ind.shape[0]
        gold_arr = _gold_vector(gold, n_rows)
    else:
        ind_host = np.ascontiguousarray(np.asarray(nn_ind), dtype=np.int64)
        if ind_host.ndim != 2:
            raise ValueError("nn_ind must be a 2D neighbour index matrix")
        gold_arr = _gold_vector(gold, ind_host.shape[0])
    if isinstance(nn_ind, N.DeviceArray):
        ctx = nn_ind.ctx
        ind_dev = nn_ind
    else:
        ctx = ctx or N.Context.get()
        ind_dev = ctx.to_device(ind_host)
    n, cols = ind_dev.shape
    hist = ctx.empty((cols + 1,), np.int64)
    N._check(ctx.lib.kz_hit_positions(ctx.handle, ind_dev.ptr, ctx.to_device(gold_arr).ptr, n, cols, hist.ptr), "kz_hit_positions")
    cum = np.cumsum(hist.numpy()[:cols])
    return {kk: (float(cum[min(kk, cols) - 1]) / len(gold) if kk >= 1 else 0.0) for kk in k}
