"""GPU box: is the sweep held back by the clock the chip keeps under load?  The same launch (ordinary fp16 kernel, 262k x 250k x 200 and
100k x 100k x 128, K' = 16) on uniform random rows and on CONSTANT rows (every row the same vector: the centred fp16 image is all
zeros -- the MFMAs multiply zeros; after the first tile no key beats the list threshold, so the candidate scan finds no event).
The scan-less diagnostic build (tools/ablate.sh 1) on random data separates what the events cost from what the data costs."""
import sys
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N

ctx = N.Context.get()
ctx.set_option("h_q64", 0)
ctx.set_option("tier_probe", 0)     # (the probe would see that constant rows cannot be certified and start the call at the split-bf16 tier: the
ctx.set_option("list_floor", 0)     #  point here is the SAME fp16 launch on three kinds of operand bits)
rng = np.random.RandomState(0)
LAUNCHES = int(sys.argv[1]) if len(sys.argv) > 1 else 4
# (constant rows tie everywhere: every row ends on the exact kernels, ~0.3 ms a row -- the query side is ONE full round of workgroups, 768 tiles)
for n_q, n_i, d in ((98_304, 250_000, 200), (98_304, 100_000, 128)):
    v = rng.rand(1, d).astype(np.float32)
    sets = {"uniform random": (rng.rand(n_q, d).astype(np.float32), rng.rand(n_i, d).astype(np.float32)),
            "constant rows (fp16 image = zeros)": (np.repeat(v, n_q, 0), np.repeat(v, n_i, 0)),
            "small integers 0..3": (rng.randint(0, 4, (n_q, d)).astype(np.float32), rng.randint(0, 4, (n_i, d)).astype(np.float32))}
    for name, (a, b) in sets.items():
        q, y = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
        ms = []
        for _ in range(LAUNCHES):
            _, _, st = N.knn(ctx, q, y, 10)
            ms.append(st["main_kernel_ms"])
        print(f"{n_q} x {n_i} x {d}  {name:36s}: main kernel {min(ms[1:]):8.3f} ms   escalated {st['n_escalated_rows']} fallback {st['n_fallback_rows']}", flush=True)
        del q, y
