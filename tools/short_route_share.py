"""GPU box: the short-list route at the size of an 8-GPU share of C3 (500k target rows x 62.5k source rows per rank, k = 50): ranges
of 48 tiles.  Time of kz_knn_dual with the route's size gate at 64 (not taken) and lower.  python3 tools/short_route_share.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N
ctx = N.Context.get()
rng = np.random.default_rng(9)
d, k = 200, 50
centres = rng.standard_normal((40, d)) * 3


def gen(kind, n):
    if kind == "uniform":
        return rng.random((n, d))
    if kind == "normal":
        return rng.standard_normal((n, d))
    if kind == "cluster by cluster":
        sizes = rng.multinomial(n, np.ones(40) / 40)
        return np.concatenate([centres[c] + 0.4 * rng.standard_normal((sizes[c], d)) for c in range(40)])
    return centres[rng.integers(0, 40, n)] + 0.4 * rng.standard_normal((n, d))


for kind in ("uniform", "normal", "cluster by cluster", "clusters shuffled"):
    a, b = gen(kind, 500000).astype(np.float32), gen(kind, 62500).astype(np.float32)
    am, bm = N.DeviceMatrix(ctx, a, "cosine"), N.DeviceMatrix(ctx, b, "cosine")
    ref = None
    for mt in (64, 32, 16, 64, 32, 16):
        ctx.set_option("dual_short_min_tiles", mt)
        ctx.sync()
        t0 = time.perf_counter()
        (xd, xi, sa), (yd, yi, sb) = N.knn_dual(ctx, am, bm, k)
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3
        if ref is None:
            ref = (xi.numpy(), yi.numpy())
        same = np.array_equal(ref[0], xi.numpy()) and np.array_equal(ref[1], yi.numpy())
        print(f"{kind:20s} min_tiles {mt:3d}: {ms:7.1f} ms dual {sa['dual']}/{sb['dual']} main {sa['main_kernel_ms']:.1f} splits {sa['n_splits']} "
              f"re-searched {sa['n_escalated_rows']}/{sb['n_escalated_rows']} same {same}", flush=True)
ctx.set_option("dual_short_min_tiles", 128)
