"""GPU box: the short-list route of kz_knn_dual on data that is NOT uniform, at a size where the route is taken by default:
time and re-searched rows with the route on and off.  python3 tools/short_route_stress.py [n] [d] [k]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
ctx = N.Context.get()
import os
for _o in os.environ.get("KZ_OPTS", "").split():
    ctx.set_option(_o.split("=")[0], float(_o.split("=")[1]))
rng = np.random.default_rng(5)


def gen(kind, n):
    if kind == "normal":
        return rng.standard_normal((n, d))
    centres = rng.standard_normal((40, d)) * 3
    if kind == "clusters, stored cluster by cluster":
        sizes = rng.multinomial(n, np.ones(40) / 40)
        return np.concatenate([centres[c] + 0.4 * rng.standard_normal((sizes[c], d)) for c in range(40)])
    if kind == "clusters, shuffled":
        return centres[rng.integers(0, 40, n)] + 0.4 * rng.standard_normal((n, d))
    if kind == "clusters of very different density":
        sc = 0.05 * 2.0 ** rng.integers(0, 6, 40)
        c = rng.integers(0, 40, n)
        return centres[c] + sc[c, None] * rng.standard_normal((n, d))
    raise ValueError(kind)


for metric in (os.environ.get("KZ_METRIC", "cosine"),):
    for kind in ("normal", "clusters, stored cluster by cluster", "clusters, shuffled", "clusters of very different density"):
        a, b = gen(kind, n).astype(np.float32), gen(kind, n + 1000).astype(np.float32)
        am, bm = N.DeviceMatrix(ctx, a, metric), N.DeviceMatrix(ctx, b, metric)
        res = {}
        for short in (0, 1, 0, 1):
            ctx.set_option("dual_short_main", short & 1)
            ctx.sync()
            t0 = time.perf_counter()
            (xd, xi, sa), (yd, yi, sb) = N.knn_dual(ctx, am, bm, k)
            ctx.sync()
            ms = (time.perf_counter() - t0) * 1e3
            res[short] = (ms, sa, sb, xi.numpy(), yi.numpy())
        same = np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[1][4])
        for short in (0, 1):
            ms, sa, sb = res[short][:3]
            print(f"{metric:10s} {kind:38s} short={short}: {ms:7.1f} ms  dual {sa['dual']}/{sb['dual']} main {sa['main_kernel_ms']:.1f} splits {sa['n_splits']} "
                  f"re-searched {sa['n_escalated_rows']}/{sb['n_escalated_rows']} exact-fallback {sa['n_fallback_rows']}/{sb['n_fallback_rows']} ev/row {sb['n_events'] / len(b):.0f}", flush=True)
        print("   same neighbours:", same, flush=True)
ctx.set_option("dual_short_main", 1)
