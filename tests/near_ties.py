"""Near-tie probe (test infrastructure; fixture tests/golden/near_ties.npz from tools/gen_near_ties.py): 1024 queries whose two
nearest index rows differ in exact squared distance by 1/64 .. 16 ulps of s = |q|^2 + |y|^2 -- the scale at which both the
reference's dgemm expansion and the device's float64 re-rank round.  Reports, per gap bucket, how often the device orders the pair
as the reference did, and how often each of them agrees with exact arithmetic ("residual risk")."""
from pathlib import Path

import numpy as np

BUCKETS = ((0.0, 1 / 16), (1 / 16, 1 / 4), (1 / 4, 1.0), (1.0, 2.0), (2.0, 1e30))


def run_probe(ctx=None):
    from kiez_amd import _native as N
    ctx = ctx or N.Context.get()
    fx = np.load(Path(__file__).resolve().parent / "golden" / "near_ties.npz")
    q, index, gap, exact, ref = fx["query"], fx["index"], fx["gap_ulps"], fx["exact_nearer"], fx["ref_ind"]
    dist, ind, st = N.knn(ctx, N.DeviceMatrix(ctx, q, "sqeuclidean"), N.DeviceMatrix(ctx, index, "sqeuclidean"), 2)
    ind = ind.numpy()
    pair_ok = bool((np.sort(ind, axis=1) == np.sort(ref, axis=1)).all())    # the same two rows for every query, whatever their order
    out = {"pairs": int(len(gap)), "same_two_rows_for_every_query": pair_ok, "gap_unit": "ulps of |q|^2 + |y|^2 (float64)", "buckets": []}
    for lo, hi in BUCKETS:
        sel = (gap >= lo) & (gap < hi)
        n = int(sel.sum())
        out["buckets"].append({"gap_ulps": [lo, None if hi > 1e29 else hi], "pairs": n,
                               "device_orders_as_reference": float((ind[sel, 0] == ref[sel, 0]).mean()) if n else None,
                               "device_orders_as_exact_arithmetic": float((ind[sel, 0] == exact[sel]).mean()) if n else None,
                               "reference_orders_as_exact_arithmetic": float((ref[sel, 0] == exact[sel]).mean()) if n else None})
    out["max_err_ratio"] = st["max_err_ratio"]
    return out
