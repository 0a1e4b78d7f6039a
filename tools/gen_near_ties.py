#!/usr/bin/env python3
"""Adversarial fixture for the ORDER of float64 near-ties ("residual risk").

The certified candidate sets make the device's neighbour order exact with respect to ITS float64 distance values; the reference's
values (scikit-learn's EuclideanArgKmin: |x|^2 - 2 x.y + |y|^2 with the middle term from dgemm) come from another summation order.
Two index rows whose exact squared distances to a query differ by a few ulps can therefore come out in either order -- in both
implementations.  This script builds such pairs ON PURPOSE and records what the reference does with them here:

  * 1024 queries q_i (float64, d = 64, rng.rand); index rows 2i and 2i+1 are the pair of query i: y = q_i + u (|u| = 0.5: far
    nearer than any other row) and y' = y with ONE coordinate nudged so that the exact gap |d2(q, y') - d2(q, y)| is about
    m ulps OF THE SCALE BOTH IMPLEMENTATIONS ROUND AT, s = |q|^2 + |y|^2 (the expansion cancels: d2 = 0.25 against s = 42), m spread
    from 1/64 to 16; which of the two is nearer alternates at random.  2048 more rows are random.
  * per pair the EXACT gap in ulps of s (rational arithmetic on the float64 inputs) and the exactly nearer row;
  * the reference's answer: kiez SklearnNN(metric="sqeuclidean").kneighbors(k=2) run through tools/ref_loader.py -- indices and
    distances as scikit-learn / the BLAS of this container produce them.

-> tests/golden/near_ties.npz (tests/test_gpu_near_ties.py, bench.py's `fp64_order_probe`).
Run in the build container only (needs /root/reference)."""
import os
import sys
import warnings
from fractions import Fraction
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from ref_loader import load_reference  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
warnings.simplefilter("ignore")


def exact_d2(a, b):
    return sum((Fraction(float(x)) - Fraction(float(y))) ** 2 for x, y in zip(a, b))


def main():
    R = load_reference()
    rng = np.random.RandomState(20261003)
    P, d = 1024, 64
    q = rng.rand(P, d)
    index = rng.rand(2 * P + 2048, d)
    gap_ulps = np.zeros(P)
    nearer = np.zeros(P, dtype=np.int64)      # the exactly nearer row of the pair
    target_m = 2.0 ** rng.uniform(-6, 4, P)   # 1/64 .. 16 ulps of s
    for i in range(P):
        u = rng.standard_normal(d)
        u *= 0.5 / np.linalg.norm(u)
        y = q[i] + u
        c = int(np.argmax(np.abs(u)))                       # nudge the coordinate with the largest offset: finest control of the gap
        scale = float(np.dot(q[i], q[i]) + np.dot(y, y))
        delta = target_m[i] * np.spacing(scale) / (2.0 * abs(y[c] - q[i][c]))
        y2 = y.copy()
        y2[c] = y[c] + (delta if (y[c] - q[i][c]) > 0 else -delta) * (1 if rng.rand() < 0.5 else -1)
        if y2[c] == y[c]:
            y2[c] = np.nextafter(y[c], 2.0)
        e1, e2 = exact_d2(q[i], y), exact_d2(q[i], y2)
        first, second = (y, y2) if rng.rand() < 0.5 else (y2, y)   # which index row holds which
        index[2 * i], index[2 * i + 1] = first, second
        ef, es = (e1, e2) if first is y else (e2, e1)
        gap_ulps[i] = float(abs(ef - es) / Fraction(float(np.spacing(scale))))
        nearer[i] = 2 * i if ef < es else (2 * i + 1 if es < ef else -1)
    nn = R.SklearnNN(n_candidates=2, metric="sqeuclidean", algorithm="brute")
    nn.fit(q, index)
    dist, ind = nn.kneighbors(k=2, return_distance=True)
    assert all(set(ind[i]) == {2 * i, 2 * i + 1} for i in range(P)), "a pair is not its query's two nearest rows"
    np.savez_compressed(OUT / "near_ties.npz", query=q, index=index, gap_ulps=gap_ulps, exact_nearer=nearer,
                        ref_ind=ind.astype(np.int64), ref_dist=dist)
    agree_exact = ind[:, 0] == nearer
    for lo, hi in ((0, 1 / 16), (1 / 16, 1 / 4), (1 / 4, 1), (1, 4), (4, 1e9)):
        sel = (gap_ulps >= lo) & (gap_ulps < hi)
        print(f"gap [{lo}, {hi}) ulps: {int(sel.sum()):4d} pairs, reference orders {float(agree_exact[sel].mean()) * 100 if sel.any() else float('nan'):5.1f} % of them as exact arithmetic does")
    print("wrote", OUT / "near_ties.npz")


if __name__ == "__main__":
    if os.environ.get("OMP_NUM_THREADS") is None:
        os.environ["OMP_NUM_THREADS"] = "8"
    main()
