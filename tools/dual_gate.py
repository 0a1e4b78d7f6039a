"""GPU box: where the shared sweep (kz_knn_dual) starts to pay -- two ordinary kz_knn calls against the forced shared sweep over a
grid of mid-size shapes and strides (round 4: the gate's constants dated from before the threshold rank and the seeded lists).
    python3 tools/dual_gate.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from kiez_amd import _native as N


def best_of(fn, ctx, reps=4):
    best = 1e9
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter(); fn(); ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def main():
    ctx = N.Context.get()
    rng = np.random.default_rng(1)
    shapes = [(30_000, 30_000, 128, 10), (50_000, 50_000, 128, 10), (70_000, 70_000, 128, 10), (100_000, 100_000, 128, 10),
              (100_000, 100_000, 64, 10), (100_000, 100_000, 300, 10), (150_000, 60_000, 128, 10), (200_000, 50_000, 200, 10),
              (100_000, 100_000, 128, 50), (60_000, 60_000, 200, 50), (150_000, 150_000, 128, 10), (40_000, 200_000, 128, 10)]
    for na, nb, d, k in shapes:
        a = rng.random((na, d), dtype=np.float32)
        b = rng.random((nb, d), dtype=np.float32)
        am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
        ctx.set_option("dual_force", 0); ctx.set_option("dual_stride", 0)
        t_sep = best_of(lambda: N.knn_dual(ctx, am, bm, k), ctx)       # (stride 0: always two ordinary searches)
        out = []
        for stride in (1, 3, 4, 5, 6, 8):
            ctx.set_option("dual_force", 1); ctx.set_option("dual_stride", stride)
            st = {}
            def f():
                (_, _, sa), (_, _, sb) = N.knn_dual(ctx, am, bm, k)
                st["dual"] = sa["dual"]
            t = best_of(f, ctx)
            out.append(f"s{stride}{'' if st['dual'] else '(sep)'} {t:.2f}")
        ctx.set_option("dual_force", 0); ctx.set_option("dual_stride", 1)
        st = {}
        def g():
            (_, _, sa), (_, _, sb) = N.knn_dual(ctx, am, bm, k)
            st["dual"] = sa["dual"]
        t_auto = best_of(g, ctx)
        t_model = 2.0 * na * nb * (((d + 15) // 16) * 16) / 1e12
        print(f"{na}x{nb}x{d} k={k}: T {t_model:.2f} | two searches {t_sep:.2f} ms | forced: {' '.join(out)} | gate: {'shared' if st['dual'] else 'two'} {t_auto:.2f}", flush=True)
        del am, bm
    ctx.set_option("dual_force", 0); ctx.set_option("dual_stride", 1)


if __name__ == "__main__":
    main()
