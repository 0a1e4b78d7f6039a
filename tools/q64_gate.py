"""GPU box: where the 64-queries-per-wave kernel pays -- main-kernel time of kz_knn (ordinary fp16 kernel, K' = 16) with h_q64 = 0 / 1
over the number of query rows, for three slice counts.   python3 tools/q64_gate.py
(The third column of profiles/r04_ablation.md section 2, "copies issued late", came from a knob that was removed after measuring.)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N

ctx = N.Context.get()
rng = np.random.RandomState(0)
for d, n_i in ((200, 250_000), (128, 100_000), (64, 200_000)):
    y = N.DeviceMatrix(ctx, rng.rand(n_i, d).astype(np.float32), "euclidean")
    for n_q in (100_000, 200_000, 400_000, 1_000_000):
        q = N.DeviceMatrix(ctx, rng.rand(n_q, d).astype(np.float32), "euclidean")
        res = {}
        for rnd in range(3):
            for name, opts in (("q32", {"h_q64": 0}), ("q64", {"h_q64": 1})):
                for k_, v_ in opts.items():
                    ctx.set_option(k_, v_)
                _, ind, st = N.knn(ctx, q, y, 10)
                res.setdefault(name, []).append(st["main_kernel_ms"])
                if rnd == 0:
                    res.setdefault("ind", []).append(ind.numpy())
        same = all(np.array_equal(res["ind"][0], x) for x in res["ind"][1:])
        print(f"d={d:4d} n_i={n_i:7d} n_q={n_q:8d} ({(n_q + 127) // 128:5d} tiles): " + "  ".join(f"{n} {min(res[n]):8.3f}" for n in ("q32", "q64")) + f"  same={same}", flush=True)
        del q
ctx.set_option("h_q64", 2)
