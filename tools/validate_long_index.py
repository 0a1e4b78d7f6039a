#!/usr/bin/env python3
"""Full-length validation of the fused kernels against the oracle (GPU box): long index sweeps exercise the LDS-DMA ring
and the candidate lists the way the small parity tests cannot.   python3 tools/validate_long_index.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from kiez_amd import _native as N  # noqa: E402
from oracle import kiez_oracle as O  # noqa: E402  (checker only)

CASES = [  # n_index, d, metric, k, n_query
    (1_000_000, 300, "euclidean", 10, 3000),
    (500_000, 200, "cosine", 50, 2000),
    (400_000, 128, "sqeuclidean", 10, 3000),
    (300_000, 384, "euclidean", 100, 1000),
]


def main():
    ctx = N.Context.get()
    bad = 0
    for n_i, d, metric, k, n_q in CASES:
        rng = np.random.default_rng(n_i + d)
        t = rng.random((n_i, d), dtype=np.float32)
        s = rng.random((n_q, d), dtype=np.float32)
        if metric == "cosine":
            t, s = t.astype(np.float64), s.astype(np.float64)
        ym, qm = N.DeviceMatrix(ctx, t, metric), N.DeviceMatrix(ctx, s, metric)
        worst = 0.0
        for _ in range(3):
            dd, ii, st = N.knn(ctx, qm, ym, k)
            worst = max(worst, st["max_err_ratio"])
        t0 = time.time()
        od, oi = O.knn_exact(s, t, k, O.canonical_metric(metric))
        same = int((ii.numpy() == oi).all(axis=1).sum())
        ok = same == n_q and worst < 0.5
        bad += 0 if ok else 1
        print("ok " if ok else "BAD", f"{n_i}x{d} {metric} k={k}: rows identical {same}/{n_q}, max_err_ratio {worst:.4f},",
              f"tier {st['first_pass']}, escalated {st['n_escalated_rows']}, fallback {st['n_fallback_rows']},",
              f"kernel {st['main_kernel_ms']:.2f} ms (oracle {time.time() - t0:.1f} s)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
