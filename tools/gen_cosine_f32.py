#!/usr/bin/env python3
"""Fixture for the ONE convention where this build deliberately does not follow the reference's arithmetic: cosine + float32 inputs.

The reference's cosine path evaluates float32 inputs in float32 (sklearn normalises in float32 and calls sgemm,
sklearn/metrics/pairwise.py:1166-1175, 1728-1736; kiez/neighbors/exact/sklearn_nearest_neighbors.py:96-101), so its neighbour ORDER
is decided by sgemm rounding wherever two candidates are closer than ~1e-7 relative.  The device (and the oracle) treat float32 inputs
as their exact float64 casts (SURVEY.md 8c caution 2).  This script records what the reference itself returns on
float32 inputs -- C3's shape at fixture scale: 2000 x 1500 x 200, cosine, k = 50, rng.rand -- so that a test can COUNT the rows the
two conventions order differently and show that every difference lies inside a group of candidates whose exact float64 distances
agree to 1e-6 relative (tests/cosine_f32.py, tests/test_gpu_cosine_f32.py, bench.py `cosine_f32_probe`).

Inputs are regenerated from the seed by the test (RandomState is stable across numpy versions); the fixture holds the reference's
outputs only: ind as int16, dist as float32.        -> tests/golden/cosine_f32_sgemm.npz        Build container only."""
import sys
import warnings
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from ref_loader import load_reference  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
SEED, N_S, N_T, D, K = 20261004, 2000, 1500, 200, 50


def inputs():
    rng = np.random.RandomState(SEED)
    return rng.rand(N_S, D).astype(np.float32), rng.rand(N_T, D).astype(np.float32)


def main():
    warnings.simplefilter("ignore")
    R = load_reference()
    s, t = inputs()
    hub = R.NoHubnessReduction(nn_algo=R.SklearnNN(n_candidates=K, metric="cosine", algorithm="brute"))
    hub.fit(s, t)
    d32, i32 = hub.kneighbors(K)                       # the reference on the float32 inputs (sgemm order)
    assert d32.dtype == np.float32 and i32.max() < 32768
    hub64 = R.NoHubnessReduction(nn_algo=R.SklearnNN(n_candidates=K, metric="cosine", algorithm="brute"))
    hub64.fit(s.astype(np.float64), t.astype(np.float64))
    d64, i64 = hub64.kneighbors(K)                     # the reference on the exact float64 casts (the convention of this build)
    differ = int((i32 != i64).any(axis=1).sum())
    np.savez_compressed(OUT / "cosine_f32_sgemm.npz", seed=np.int64(SEED), shape=np.array([N_S, N_T, D, K]),
                        ref_f32_ind=i32.astype(np.int16), ref_f32_dist=d32, ref_f64cast_ind=i64.astype(np.int16),
                        rows_reference_itself_orders_differently=np.int64(differ))
    print(f"wrote cosine_f32_sgemm.npz: the reference on float32 inputs and on their float64 casts orders {differ} of {N_S} rows differently")


if __name__ == "__main__":
    main()
