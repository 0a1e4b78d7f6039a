"""GPU box: kz_knn_dual against two ordinary kz_knn calls (bit-exact), with timings.   python3 tools/dual_check.py [cases...]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N

CASES = {
    "small": (20000, 6000, 64, 10, "euclidean"),
    "c2": (100000, 100000, 128, 10, "euclidean"),
    "odd": (50000, 30011, 72, 5, "sqeuclidean"),
    "cos": (60000, 40000, 200, 50, "cosine"),
    "k26": (40000, 20000, 48, 26, "euclidean"),
    "ns": (1000000, 250000, 200, 10, "euclidean"),
    "c3": (500000, 500000, 200, 50, "cosine"),
    "c3e": (500000, 500000, 200, 50, "euclidean"),
}

def run(name):
    na, nb, d, k, metric = CASES[name]
    rng = np.random.default_rng(len(name) + na)
    if name in ("cos", "c3", "c3e"):
        a = rng.standard_normal((na, d), dtype=np.float32)
        b = rng.standard_normal((nb, d), dtype=np.float32)
    else:
        a = rng.random((na, d), dtype=np.float32)
        b = rng.random((nb, d), dtype=np.float32)
    ctx = N.Context.get()
    ctx.set_option("dual_force", 1)
    import os
    if os.environ.get("DUAL_STRIDE"):
        ctx.set_option("dual_stride", int(os.environ["DUAL_STRIDE"]))
    am, bm = N.DeviceMatrix(ctx, a, metric), N.DeviceMatrix(ctx, b, metric)
    # warm (images, pools)
    N.knn(ctx, am, bm, k); N.knn(ctx, bm, am, k)
    ctx.sync(); t0 = time.perf_counter()
    dab, iab, s1 = N.knn(ctx, am, bm, k)
    dba, iba, s2 = N.knn(ctx, bm, am, k)
    ctx.sync(); t_sep = time.perf_counter() - t0
    N.knn_dual(ctx, am, bm, k)
    ctx.sync(); t0 = time.perf_counter()
    (xd, xi, sa), (yd, yi, sb) = N.knn_dual(ctx, am, bm, k)
    ctx.sync(); t_dual = time.perf_counter() - t0
    ok_ab = np.array_equal(iab.numpy(), xi.numpy()) and np.array_equal(dab.numpy(), xd.numpy())
    ok_ba = np.array_equal(iba.numpy(), yi.numpy()) and np.array_equal(dba.numpy(), yd.numpy())
    nbad = int((iba.numpy() != yi.numpy()).any(axis=1).sum())
    print(f"{name}: a->b {'OK' if ok_ab else 'BAD'}  b->a {'OK' if ok_ba else 'BAD (%d rows)' % nbad} | separate {t_sep*1e3:.1f} ms "
          f"(main {s1['main_kernel_ms']:.1f}+{s2['main_kernel_ms']:.1f}) dual {t_dual*1e3:.1f} ms (sweep {sa['main_kernel_ms']:.1f}, fin {sa['finalize_ms']:.1f}; "
          f"reverse: sample+scatter+select {sb['main_kernel_ms']:.1f}, fin {sb['finalize_ms']:.1f}, fb {sb['fallback_ms']:.1f}) "
          f"dual={sa['dual']}/{sb['dual']} events/row {sb['n_events']/nb:.1f} logged/row {sb['n_logged_groups']/nb:.1f} overflow {sb['n_overflow_rows']} esc {sb['n_escalated_rows']} "
          f"err {sa['max_err_ratio']:.3f}/{sb['max_err_ratio']:.3f} blocks {s1['n_blocks']}/{sa['n_blocks']} splits {s1['n_splits']}/{sa['n_splits']}", flush=True)
    return ok_ab and ok_ba

if __name__ == "__main__":
    names = sys.argv[1:] or ["small", "odd", "k26", "cos", "c2"]
    bad = [n for n in names if not run(n)]
    print("FAILED:" if bad else "all ok", bad)
    sys.exit(1 if bad else 0)
