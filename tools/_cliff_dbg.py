import sys, time
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N
sys.argv = [sys.argv[0]]
exec(open("tools/cliff_probe.py").read().split("KINDS = [")[0].split("ctx = N.Context.get()")[1])
ctx = N.Context.get()
def run(n, d, k, metric, kind, opts):
    for o, v in opts.items(): ctx.set_option(o, v)
    rng = np.random.default_rng(11)
    a, b = gen(kind, n, d, rng).astype(np.float32), gen(kind, n + 1000, d, rng).astype(np.float32)
    am, bm = N.DeviceMatrix(ctx, a, metric), N.DeviceMatrix(ctx, b, metric)
    best = None
    for _ in range(2):
        ctx.sync(); t0 = time.perf_counter()
        (xd, xi, sa), (yd, yi, sb) = N.knn_dual(ctx, am, bm, k)
        ctx.sync(); ms = (time.perf_counter() - t0) * 1e3
        best = ms if best is None or ms < best else best
    print(f"{n} d={d} k={k} {metric} {kind[:30]:30s} {opts}: {best:8.1f} ms", flush=True)
    for s in (sa, sb):
        print("    ", {x: (round(s[x], 2) if isinstance(s[x], float) else s[x]) for x in ("main_kernel_ms", "finalize_ms", "fallback_ms", "n_fallback_rows", "list_len", "n_splits", "first_pass", "n_escalated_rows", "dual", "n_first_pass_fail", "wide_lists", "probe_ms")}, flush=True)
    for o in opts: ctx.set_option(o, {"precision": 0, "tier_probe": 1024, "wide_lists": 32}.get(o, 0))
K1 = "40 tight clusters, shuffled"
run(200_000, 200, 10, "euclidean", K1, {})
run(200_000, 200, 10, "euclidean", K1, {"precision": 2})
run(300_000, 64, 50, "cosine", K1, {})
run(300_000, 64, 50, "cosine", K1, {"precision": 2})
