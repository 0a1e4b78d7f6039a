"""GPU box: seeded lists on ORDINARY searches (kz_knn, >= 5e10 pairs: the tier probe's results feed the floor model).
Same process, interleaved: list_floor = 0 / 1 over a few shapes; results must be identical, times are printed."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kiez_amd import _native as N  # noqa: E402


def run(ctx, qm, ym, k, reps=3):
    best, out = 1e9, None
    for _ in range(reps):
        t0 = time.perf_counter()
        d, i, st = N.knn(ctx, qm, ym, k)
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3)
        out = (d, i, st)
    return best, out


def main():
    ctx = N.Context.get()
    rng = np.random.RandomState(3)
    shapes = [(300_000, 300_000, 128, 10, "euclidean", "uniform"), (200_000, 400_000, 200, 50, "cosine", "uniform"),
              (300_000, 300_000, 96, 10, "euclidean", "clustered"), (250_000, 1_000_000, 200, 10, "euclidean", "uniform")]
    for nq, ny, d, k, metric, kind in shapes:
        if kind == "uniform":
            q = rng.rand(nq, d).astype(np.float32)
            y = rng.rand(ny, d).astype(np.float32)
        else:
            cen = rng.randn(64, d).astype(np.float32) * 2
            sc = (0.05 + 0.5 * rng.rand(64, 1)).astype(np.float32)
            cq, cy = rng.randint(0, 64, nq), rng.randint(0, 64, ny)
            q = cen[cq] + sc[cq] * rng.randn(nq, d).astype(np.float32)
            y = cen[cy] + sc[cy] * rng.randn(ny, d).astype(np.float32)
        qm, ym = N.DeviceMatrix(ctx, q, metric), N.DeviceMatrix(ctx, y, metric)
        res = {}
        for rnd in range(2):
            for fl in (0, 1):
                ctx.set_option("list_floor", fl)
                ms, (dd, ii, st) = run(ctx, qm, ym, k)
                res.setdefault(fl, []).append((ms, st["main_kernel_ms"], st["n_escalated_rows"], st["first_pass"]))
                if rnd == 0:
                    res[("out", fl)] = (dd.numpy(), ii.numpy())
        same = np.array_equal(res[("out", 0)][1], res[("out", 1)][1]) and np.array_equal(res[("out", 0)][0], res[("out", 1)][0])
        for fl in (0, 1):
            print(f"{nq}x{ny}x{d} k={k} {metric} {kind}: list_floor={fl} call ms {[round(r[0], 2) for r in res[fl]]} main {[round(r[1], 2) for r in res[fl]]}"
                  f" esc {res[fl][-1][2]} tier {res[fl][-1][3]}" + (f" identical={same}" if fl else ""), flush=True)
        del qm, ym
        ctx.trim()
    ctx.set_option("list_floor", 0)


if __name__ == "__main__":
    main()
