"""GPU box: how much slower is the search on data that is NOT uniform?  kz_knn_dual (both directions of a hubness-reduced fit +
kneighbors) on six kinds of data at a few shapes -- time, ratio to uniform rows of the same shape, and the route the call took
(first-pass tier, wide lists, rows re-searched, rows on the exact kernels).  Round 4's verdict: clustered data must not fall off a
cliff (hard: 5 x uniform then).      python3 tools/cliff_probe.py [n d k metric]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from kiez_amd import _native as N  # noqa: E402

ctx = N.Context.get()
TIER = {0: "f32", 1: "bf16x2", 2: "fp16"}


def gen(kind, n, d, rng):
    if kind == "uniform":
        return rng.random((n, d))
    if kind == "normal":
        return rng.standard_normal((n, d))
    centres = np.random.default_rng(5).standard_normal((40, d)) * 3      # (the same centres on both sides)
    if kind == "40 tight clusters, stored cluster by cluster":
        sizes = rng.multinomial(n, np.ones(40) / 40)
        return np.concatenate([centres[c] + 0.4 * rng.standard_normal((sizes[c], d)) for c in range(40)])
    if kind == "40 tight clusters, shuffled":
        return centres[rng.integers(0, 40, n)] + 0.4 * rng.standard_normal((n, d))
    if kind == "clusters of very different density":
        sc = 0.05 * 2.0 ** np.random.default_rng(6).integers(0, 6, 40)
        c = rng.integers(0, 40, n)
        return centres[c] + sc[c, None] * rng.standard_normal((n, d))
    if kind == "256-component mixture, L2-normalised":
        cc = np.random.default_rng(7).standard_normal((256, d))
        x = cc[rng.integers(0, 256, n)] + 0.35 * rng.standard_normal((n, d))
        return x / np.sqrt((x * x).sum(axis=1, keepdims=True))
    raise ValueError(kind)


KINDS = ["uniform", "normal", "40 tight clusters, stored cluster by cluster", "40 tight clusters, shuffled", "clusters of very different density",
         "256-component mixture, L2-normalised"]
SHAPES = ((300_000, 64, 50, "cosine"), (300_000, 96, 10, "euclidean"), (200_000, 200, 10, "euclidean"))
if len(sys.argv) > 4:      # one shape from the command line: n d k metric
    SHAPES = ((int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]),)
for n, d, k, metric in SHAPES:
    base = None
    for kind in KINDS:
        rng = np.random.default_rng(11)
        a, b = gen(kind, n, d, rng).astype(np.float32), gen(kind, n + 1000, d, rng).astype(np.float32)
        am, bm = N.DeviceMatrix(ctx, a, metric), N.DeviceMatrix(ctx, b, metric)
        best = None
        for _ in range(3):
            ctx.sync()
            t0 = time.perf_counter()
            (xd, xi, sa), (yd, yi, sb) = N.knn_dual(ctx, am, bm, k)
            ctx.sync()
            ms = (time.perf_counter() - t0) * 1e3
            best = ms if best is None or ms < best else best
        base = best if base is None else base
        print(f"{n // 1000}k x {n // 1000 + 1}k x {d} k={k} {metric:9s} {kind:46s} {best:7.1f} ms  x{best / base:4.2f}  shared {sa['dual']}  first pass {TIER[sa['first_pass']]}"
              f"  wide lists {sa['wide_lists']}/{sb['wide_lists']}  re-searched {sa['n_escalated_rows']}/{sb['n_escalated_rows']}  exact {sa['n_fallback_rows']}/{sb['n_fallback_rows']}", flush=True)
        del am, bm
