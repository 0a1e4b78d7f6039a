"""GPU box: what the speculative exact re-search of a handful of rows costs (kz_knn.hip "SPECULATIVE RESCUE") -- every row forced to fail
(eps_scale 1e30), 4 / 16 query rows against large indexes, with the cooperative exact kernel (exact_rows 1) and the one-pair-per-lane
kernel (2).    python3 tools/spec_time.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from kiez_amd import _native as N
ctx = N.Context.get()
rng = np.random.default_rng(0)
for n_i, d, metric in ((500_000, 200, "cosine"), (500_000, 200, "euclidean"), (1_000_000, 200, "euclidean"), (100_000, 128, "euclidean")):
    y = rng.random((n_i, d), dtype=np.float32)
    ym = N.DeviceMatrix(ctx, y, metric)
    for nq in (4, 16):
        q = rng.random((nq, d), dtype=np.float32)
        qm = N.DeviceMatrix(ctx, q, metric)
        ctx.set_option("eps_scale", 1e30)
        for er in (1, 2):
            ctx.set_option("exact_rows", er)
            best = 1e9
            for _ in range(4):
                ctx.sync(); t0 = time.perf_counter(); _, _, st = N.knn(ctx, qm, ym, 10); ctx.sync()
                best = min(best, (time.perf_counter() - t0) * 1e3)
            print(f"{n_i} x {d} {metric} rows {nq} exact_rows {er}: call {best:.3f} ms  fallback_ms {st['fallback_ms']:.3f} spec {st['n_spec_rows']}", flush=True)
        ctx.set_option("eps_scale", 1.0); ctx.set_option("exact_rows", 2)
