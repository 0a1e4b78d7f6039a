"""cosine + float32 inputs: the one convention where this build does not follow the reference's ARITHMETIC (test infrastructure;
fixture tests/golden/cosine_f32_sgemm.npz from tools/gen_cosine_f32.py).

The reference evaluates float32 inputs in float32 (normalise, sgemm, 1 - S: sklearn/metrics/pairwise.py:1166-1175, 1728-1736), so
where two candidates of a query are closer than float32 resolves, its order is sgemm's rounding.  The device treats float32 inputs
as their exact float64 casts (SURVEY.md 8c caution 2).  `run_probe` COUNTS what that costs on C3's shape at
fixture scale (2000 x 1500 x 200, k = 50): rows ordered exactly as the reference-on-float32 did; rows that differ only by
permutations inside groups of candidates whose exact float64 distances agree to 5e-6 relative (float32 cannot tell them apart);
rows that differ otherwise (expected: none)."""
from pathlib import Path

import numpy as np

# float32 cannot tell two candidates apart whose distances (1 - cos ~ 0.25 on rng.rand data) differ by less than sgemm's error on
# S ~ 0.75 accumulated over d = 200 terms: a few 1e-7 absolute = a few 1e-6 relative to the distance.  Measured on this fixture
# (tools/gen_cosine_f32.py): the 32 rows the reference itself orders differently between float32 inputs and their float64 casts
# differ by at most 2.6e-6 relative in exact distance, position by position.
TIE_RTOL = 5e-6


def inputs(fx):
    n_s, n_t, d, _ = (int(v) for v in fx["shape"])
    rng = np.random.RandomState(int(fx["seed"]))
    return rng.rand(n_s, d).astype(np.float32), rng.rand(n_t, d).astype(np.float32)


def classify(ref_ind, got_ind, exact_dist_of):
    """-> (identical, tie_permutation_only, other): per row, `exact_dist_of(row, ids)` = exact float64 distances of target ids."""
    identical = tie_only = other = 0
    for r in range(len(ref_ind)):
        a, b = ref_ind[r], got_ind[r]
        if np.array_equal(a, b):
            identical += 1
            continue
        da, db = exact_dist_of(r, a), exact_dist_of(r, b)
        # the same candidates up to what lies within the tolerance of the k-th distance, and position by position distances that
        # agree to the tolerance: the two orders differ only inside near-tie groups
        pos_ok = np.all(np.abs(da - db) <= TIE_RTOL * np.maximum(np.abs(da), np.abs(db)))
        sym = np.setxor1d(a, b)
        edge = max(da[-1], db[-1])
        set_ok = len(sym) == 0 or np.all(np.abs(exact_dist_of(r, sym) - edge) <= TIE_RTOL * edge)
        if pos_ok and set_ok:
            tie_only += 1
        else:
            other += 1
    return identical, tie_only, other


def run_probe(ctx=None):
    from kiez_amd import _native as N
    ctx = ctx or N.Context.get()
    fx = np.load(Path(__file__).resolve().parent / "golden" / "cosine_f32_sgemm.npz")
    s, t = inputs(fx)
    k = int(fx["shape"][3])
    dist, ind, st = N.knn(ctx, N.DeviceMatrix(ctx, s, "cosine"), N.DeviceMatrix(ctx, t, "cosine"), k)
    got = ind.numpy()
    s64, t64 = s.astype(np.float64), t.astype(np.float64)
    sn, tn = s64 / np.sqrt((s64 * s64).sum(1))[:, None], t64 / np.sqrt((t64 * t64).sum(1))[:, None]

    def exact(r, ids):
        return np.clip(1.0 - tn[np.asarray(ids, dtype=np.int64)] @ sn[r], 0.0, 2.0)
    ref32 = fx["ref_f32_ind"].astype(np.int64)
    ref64 = fx["ref_f64cast_ind"].astype(np.int64)
    ident, tie_only, other = classify(ref32, got, exact)
    return {"shape": [int(v) for v in fx["shape"]], "metric": "cosine", "inputs": "float32",
            "rows": int(len(got)), "rows_ordered_as_reference_on_float32": ident,
            "rows_differing_only_inside_near_tie_groups": tie_only, "rows_differing_otherwise": other,
            "tie_rtol": TIE_RTOL,
            "rows_identical_to_reference_on_float64_casts": int((got == ref64).all(axis=1).sum()),
            "rows_reference_itself_orders_differently_f32_vs_f64cast": int(fx["rows_reference_itself_orders_differently"]),
            "max_err_ratio": st["max_err_ratio"]}
